#!/bin/bash
# Copies the summaries of gpurun_out/<tag>/ (tools/collect_profiles.sh + the rehearsal benches) into profiles/ as r<NN>_*.
# usage: tools/install_profiles.sh <tag> <round prefix, e.g. r02>
set -e
T=gpurun_out/$1; R=$2
# profiles/<round>_pmc is REPLACED: nothing else may live there (the co-execution counter passes of tools/pmc_coexec.sh have their own
# directory, profiles/<round>_coexec -- an earlier install wiped them when they shared this one)
[ -d $T/pmc ] || { echo "$T/pmc missing: nothing installed"; exit 1; }
rm -rf profiles/${R}_pmc && mkdir -p profiles/${R}_pmc && cp $T/pmc/* profiles/${R}_pmc/
sed -i "s#(pmc);#(profiles/${R}_pmc);#" profiles/${R}_pmc/freq_lstm_traffic.json
cp $T/bench.json profiles/${R}_bench.json
cp $T/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
cp $T/kernel_stats.csv profiles/${R}_kernel_stats.csv
cp $T/per_launch.txt profiles/${R}_per_launch.txt
cp $T/trace/run_kernel_trace.csv profiles/${R}_kernel_trace.csv
[ -f $T/bench_8khz.json ] && cp $T/bench_8khz.json profiles/${R}_bench_8khz.json
[ -f $T/bench_config5.json ] && cp $T/bench_config5.json profiles/${R}_bench_config5_rehearsal.json
[ -f $T/bench_mesh_stage.json ] && cp $T/bench_mesh_stage.json profiles/${R}_bench_mesh_stage.json
# precision sweep (tests/test_gpu_precision.py): the fixture table and, from round 5, the wide table (other weight dynamics, the 10 s
# reference fixture, full size) in ONE file; bench.py reports its worst_case_dgrad
python3 - <<PY
import json, os
out = {}
if os.path.exists("gpurun_out/precision_modes.json"):
    out = json.load(open("gpurun_out/precision_modes.json"))
if os.path.exists("gpurun_out/precision_modes_wide.json"):
    w = json.load(open("gpurun_out/precision_modes_wide.json"))
    out.update({"worst_case_dgrad": w["worst_case_dgrad"], "bounds_asserted": w["bounds_asserted"], "cases": w["cases"]})
if out:
    json.dump(out, open("profiles/${R}_precision_modes.json", "w"), indent=1)
PY
ls -la profiles/${R}_* | head -20
# round 6: the per-mode attention counter records of tools/pmc_precision.sh <tag>_prec (bench.py: mixed_precision.<mode>.mfma_util_pct_counters)
if [ -d ${T}_prec ]; then
  cp ${T}_prec/attention_mfma_*.json profiles/${R}_pmc/ 2>/dev/null || true
  [ -f ${T}_prec/summary.txt ] && cp ${T}_prec/summary.txt profiles/${R}_pmc_precision_modes.txt
fi
# round 3: the secondary lines come from tools/collect_extras.sh <tag2>; pass it as the third argument
if [ -n "$3" ]; then
  X=gpurun_out/$3
  for f in bench_final bench_steps20 bench_8khz bench_mesh_stage bench_1x2s bench_1x10s rehearsal_nccl_w1_dgrad rehearsal_nccl_w1_expand rehearsal_nccl_w1_plain rehearsal_nccl_w1_dgrad_reserve16; do
    [ -f $X/$f.json ] && cp $X/$f.json profiles/${R}_$f.json
  done
  [ -f $X/bench_config5.json ] && cp $X/bench_config5.json profiles/${R}_bench_config5_rehearsal.json
fi
