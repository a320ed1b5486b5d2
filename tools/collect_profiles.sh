# Collects the judged measurements in ONE gpurun call: bench line, rocprofv3 --kernel-trace --stats of the same command, and
# three SEPARATE --pmc passes (never combined with tracing).  Usage: bash tools/collect_profiles.sh <tag> [extra bench args]
# Output: gpurun_out/<tag>/ ; copy the summaries into profiles/ (see profiles/README.md).
set -e
TAG=${1:-final}; shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 5 --warmup 1 "$@" > $OUT/bench.json 2> $OUT/bench.err
echo bench-done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
echo trace-done
# counters: headline path only (no column-sharing / mixed-precision passes, whose launches share symbols and grids with it),
# frequency LSTM in its hardware-dispatched form 8 (freq_lstm_v3_kernel, one workgroup per tile: the grid size tells pmc_summary.py the frame count;
# the bytes moved are the same in all four launch forms)
for C in MfmaUtil FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-mixed-precision --no-column-sharing --no-host-io --no-surface --opt freq_lstm_shape=8 "$@" > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
  echo pmc-$C-done
done
mkdir -p $OUT/pmc
for C in MfmaUtil FETCH_SIZE WRITE_SIZE; do cp $(find $OUT/pmc_$C -name "*counter_collection.csv" | head -1) $OUT/pmc/${C}_counter_collection.csv; done
python3 $R/profiles/pmc_summary.py $OUT/pmc --traffic-json $OUT/pmc/freq_lstm_traffic.json --chunks 8192,8192,3968 --frontend-json $OUT/pmc/frontend_traffic.json --frames 20352 --attention-json $OUT/pmc/attention_mfma.json > $OUT/pmc/summary.txt
python3 $R/profiles/summarize.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > $OUT/per_launch.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/pmc_MfmaUtil $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
du -sh $OUT
