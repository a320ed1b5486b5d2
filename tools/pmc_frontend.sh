# HBM counters of the spectrogram stage alone (two separate --pmc passes, never combined with tracing): tools/time_frontend.py under
# rocprofv3, for the spectral-stream form (round 5) and the two-kernel form it replaces.  Usage (GPU box): bash tools/pmc_frontend.sh <tag>
set -e
TAG=${1:-fe}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ORDER in 0 1; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_$ORDER -o run -- python3 $R/tools/time_frontend.py 1 frontend_two_kernel=$ORDER > $OUT/pmc_${C}_$ORDER.log 2>&1
    cp $(find $OUT/pmc_${C}_$ORDER -name "*counter_collection.csv" | head -1) $OUT/${C}_order$ORDER.csv
    rm -rf $OUT/pmc_${C}_$ORDER
  done
done
python3 - <<PY
import csv, collections
for order in (0, 1):
    tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for ci, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
        rows = sorted(csv.DictReader(open("$OUT/%s_order%d.csv" % (c, order))), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if name.startswith(("mel_columns", "gather_features", "share_", "mel_stream")):
                tot[name][ci] += float(r["Counter_Value"]); tot[name][2] += 1
    print("frontend_two_kernel =", order, "(0 = spectral stream, 1 = share map + mel_columns + gather_features; 3 calls of the stage: 20,352 frames each)")
    s = 0.0
    for k, (f, w, n) in sorted(tot.items()):
        calls = n // 2
        rb, wb = f * 1024 * 2 / calls / 20352, w * 1024 / calls / 20352
        s += rb + wb
        print("  %-40s read %8.1f B/frame (FETCH_SIZE x 2)  write %8.1f B/frame" % (k, rb, wb))
    print("  total %.1f KB/frame (algorithmic 99.4)" % (s / 1e3))
PY
