# Everything a round's profiles/ needs, in ONE gpurun call (about 25 minutes): tools/collect_profiles.sh <tag>, the per-mode counter passes
# (tools/pmc_precision.sh <tag>_prec: bf16x3_attention bf16x3 bf16x6) and the secondary bench lines (tools/collect_extras.sh <tag>_x).
# Install with: tools/install_profiles.sh <tag> <rNN> <tag>_x
set -e
TAG=${1:-final}
R=$GRAFT_REPO_ROOT
bash $R/tools/collect_profiles.sh $TAG > $R/gpurun_out/${TAG}_collect.log 2>&1; echo collect-done
bash $R/tools/pmc_precision.sh ${TAG}_prec bf16x3_attention bf16x3 bf16x6 > $R/gpurun_out/${TAG}_prec.log 2>&1; echo prec-done
bash $R/tools/collect_extras.sh ${TAG}_x > $R/gpurun_out/${TAG}_x.log 2>&1; echo extras-done
tail -15 $R/gpurun_out/${TAG}_x.log
