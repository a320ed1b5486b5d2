# Which HIP stream -> hardware queue assignment lets RCCL's stream run UNDER the compute stream's kernels?  World-size-1 rehearsal
# traced under three environments.  Usage (GPU box): bash tools/trace_overlap_env.sh <tag>
TAG=${1:-overlap_env}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--steps 2 --warmup 1 --no-cpu-baseline --no-mixed-precision --no-column-sharing --no-surface --no-host-io --force-gather --gather dgrad --backend nccl"
run() {
  T=$1
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/$T -o run -- python3 $R/bench.py $B $2 > $OUT/$T.json 2> $OUT/$T.err
  K=$(find $OUT/$T -name "*kernel_trace.csv" | head -1); M=$(find $OUT/$T -name "*memory_copy_trace.csv" | head -1)
  echo "== $T" > $OUT/${T}_overlap.txt
  python3 $R/profiles/overlap.py $K $M >> $OUT/${T}_overlap.txt 2>&1
  cat $OUT/${T}_overlap.txt
  rm -rf $OUT/$T
}
run default ""
export GPU_MAX_HW_QUEUES=16
run hwq16 ""
unset GPU_MAX_HW_QUEUES
export TORCH_NCCL_HIGH_PRIORITY=1
run ncclhigh ""
unset TORCH_NCCL_HIGH_PRIORITY
run sidestream "--compute-stream side"
