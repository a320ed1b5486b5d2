# Per-kernel times of the attention stage in a precision mode (rocprofv3 --kernel-trace; one short bench run).
# Usage (GPU box): bash tools/trace_attention.sh <tag> <mode> [extra bench args]
set -e
TAG=${1:-attn_trace}; MODE=${2:-bf16x3_attention}; shift 2 || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$MODE -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-mixed-precision --no-column-sharing --no-host-io --no-surface --precision $MODE "$@" > $OUT/trace_$MODE.json 2> $OUT/trace_$MODE.err
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace_$MODE/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
by = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if not any(k in name for k in ("attn", "gemm_bf16", "gemm_k4", "query")):
        continue
    k = (name, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    e = by.setdefault(k, [0, 0]); e[0] += 1; e[1] += d
print(f"mode $MODE: attention-related kernels over 3 steps (1 warm-up + 2)")
for (name, grid), (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:70]:70s} grid {grid:9d} calls {n:4d} total {t/1e6:9.3f} ms  avg {t/n/1e3:9.1f} us")
PY
rm -rf $OUT/trace_$MODE
