# VALU / MFMA co-execution counters (MI355X_MICROARCH.md "Two waves per SIMD", item 9) for the MFMA-bound kernels:
# separate rocprofv3 --pmc passes (never combined with tracing), program directly after `--`.
# Usage (on the GPU box): bash tools/pmc_coexec.sh <tag> "<label>:<bench args>" ...
set -e
TAG=${1:-coexec}; shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BASE="--steps 1 --warmup 0 --no-cpu-baseline --no-mixed-precision --no-column-sharing"
for SPEC in "$@"; do
  L=${SPEC%%:*}; ARGS=${SPEC#*:}
  P=1
  for CS in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS GRBM_GUI_ACTIVE"; do
    rocprofv3 --pmc $CS --output-format csv -d $OUT/${L}_p$P -o run -- python3 $R/bench.py $BASE $ARGS > $OUT/${L}_p$P.json 2> $OUT/${L}_p$P.err || { echo "pass $L p$P failed"; tail -5 $OUT/${L}_p$P.err; }
    F=$(find $OUT/${L}_p$P -name "*counter_collection.csv" | head -1)
    [ -n "$F" ] && cp $F $OUT/${L}_p$P.csv
    rm -rf $OUT/${L}_p$P
    echo "$L pass $P done"
    P=$((P+1))
  done
done
python3 $R/profiles/coexec_summary.py $OUT > $OUT/summary.txt || true
du -sh $OUT
