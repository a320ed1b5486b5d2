set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
echo bench-done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
echo trace-done
for C in MfmaUtil FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-mixed-precision > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
  echo pmc-$C-done
done
find $OUT -name "*.csv" | head -30
du -sh $OUT
