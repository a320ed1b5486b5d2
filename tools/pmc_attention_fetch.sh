# HBM bytes of the attention-stage kernels in a precision mode (one rocprofv3 --pmc FETCH_SIZE pass; FETCH_SIZE x 2 per MI355X_MICROARCH.md).
# Usage (GPU box): bash tools/pmc_attention_fetch.sh <tag> <mode>
set -e
TAG=${1:-attnfetch}; MODE=${2:-bf16x3_attention}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$MODE -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-mixed-precision --no-column-sharing --no-host-io --no-surface --precision $MODE > $OUT/fetch_$MODE.json 2> $OUT/fetch_$MODE.err
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_$MODE/**/*counter_collection.csv", recursive=True)[0]
by = collections.OrderedDict()
for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if not name.startswith("attn"):
        continue
    k = (name, int(r["Grid_Size"]))
    e = by.setdefault(k, [0, 0.0, 0]); e[0] += 1; e[1] += float(r["Counter_Value"]); e[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("precision $MODE, one step (20,352 frames = launch groups of 8192 / 8192 / 3968): HBM reads of the attention-stage kernels (FETCH_SIZE KiB x 2; H is 128 KiB per frame)")
for (name, grid), (n, kib, ns) in by.items():
    gb = kib * 1024 * 2 / 1e9
    print(f"  {name[:40]:40s} grid {grid:8d} calls {n:2d}  read {gb:7.3f} GB  in {ns / 1e6:7.3f} ms under the counter pass = {gb / (ns / 1e9) / 1e3:5.2f} TB/s")
PY
rm -rf $OUT/pmc_$MODE
