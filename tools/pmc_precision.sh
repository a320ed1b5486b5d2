# MfmaUtil of every kernel in a body precision mode (one rocprofv3 --pmc pass per mode, never combined with tracing).
# Usage (GPU box): bash tools/pmc_precision.sh <tag> [modes...]
set -e
TAG=${1:-pmcprec}; shift || true
MODES=${@:-bf16x3 bf16x6}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for M in $MODES; do
  rocprofv3 --pmc MfmaUtil --output-format csv -d $OUT/pmc_$M -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-mixed-precision --no-column-sharing --no-host-io --no-surface --precision $M > $OUT/pmc_$M.json 2> $OUT/pmc_$M.err
  cp $(find $OUT/pmc_$M -name "*counter_collection.csv" | head -1) $OUT/MfmaUtil_$M.csv
  rm -rf $OUT/pmc_$M
  # the attention stage of this mode (bench.py: mixed_precision.<mode>.mfma_util_pct_counters reads profiles/r*_pmc/attention_mfma_<mode>.json)
  python3 $R/profiles/pmc_summary.py --attention-only $OUT/MfmaUtil_$M.csv $OUT/attention_mfma_$M.json $M $TAG > $OUT/attention_$M.txt 2>&1 || true
  python3 - <<PY
import csv, collections
rows = sorted(csv.DictReader(open("$OUT/MfmaUtil_$M.csv")), key=lambda r: int(r["Dispatch_Id"]))
by = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if name.startswith("at::") or "rocclr" in name:
        continue
    k = (name, int(r["Grid_Size"]))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    e = by.setdefault(k, [0, 0.0, 0])
    e[0] += 1; e[1] += float(r["Counter_Value"]) * d; e[2] += d
print("precision $M: MfmaUtil per kernel and launch shape (time-weighted over its launches in one step, under the counter pass)")
print(f"{'kernel':62s} {'grid':>10s} {'calls':>5s} {'MfmaUtil %':>10s} {'ms':>9s}")
for (name, grid), (n, w, t) in sorted(by.items(), key=lambda kv: -kv[1][2]):
    if t < 20000:
        continue
    print(f"{name[:62]:62s} {grid:10d} {n:5d} {w / t:10.1f} {t / 1e6:9.3f}")
PY
done > $OUT/summary.txt 2>&1
cat $OUT/attention_*.txt >> $OUT/summary.txt 2>/dev/null || true
cat $OUT/summary.txt
