# Kernel + memory-copy trace of (1) the world-size-1 RCCL rehearsal and (2) the PCIe-inclusive twin, and for each the share of
# second-stream work that ran UNDER the main queue's kernels (profiles/overlap.py).  Usage (GPU box): bash tools/trace_overlap.sh <tag>
TAG=${1:-overlap}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--steps 2 --warmup 1 --no-cpu-baseline --no-mixed-precision --no-column-sharing --no-surface"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/w1 -o run -- python3 $R/bench.py $B --no-host-io --force-gather --gather dgrad --backend nccl > $OUT/w1.json 2> $OUT/w1.err
echo w1-done
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/hostio -o run -- python3 $R/bench.py $B > $OUT/hostio.json 2> $OUT/hostio.err
echo hostio-done
cd $R
for T in w1 hostio; do
  K=$(find $OUT/$T -name "*kernel_trace.csv" | head -1); M=$(find $OUT/$T -name "*memory_copy_trace.csv" | head -1)
  head -2 $K > $OUT/${T}_head.txt; [ -n "$M" ] && head -3 $M >> $OUT/${T}_head.txt
  python3 profiles/overlap.py $K $M > $OUT/${T}_overlap.txt 2>&1
  cat $OUT/${T}_overlap.txt
  rm -rf $OUT/$T
done
