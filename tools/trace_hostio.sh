# Catches the slow state of the PCIe-inclusive twin under a kernel + memory-copy trace and prints the main queue's idle gaps.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/hostio_trace
mkdir -p $OUT
B="--steps 3 --warmup 1 --no-cpu-baseline --no-mixed-precision --no-surface --no-column-sharing"
for i in 1 2 3 4; do
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t$i -o run -- python3 $R/bench.py $B > $OUT/t$i.json 2> $OUT/t$i.err
  python3 - <<PY
import csv, glob, json
d = json.loads(open("$OUT/t$i.json").read().strip().splitlines()[-1])
print("== run $i with_h2d_d2h ms/step", d["with_h2d_d2h"]["ms_per_step"], "device-resident", d["ms_per_step"])
k = glob.glob("$OUT/t$i/**/*kernel_trace.csv", recursive=True)[0]
m = glob.glob("$OUT/t$i/**/*memory_copy_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(k)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-50:], r["Queue_Id"] + "/" + r.get("Stream_Id", "")) for r in rows]
if m:
    ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), "copy") for r in csv.DictReader(open(m[0]))]
ev.sort()
# the host-io section: from the first copyBuffer kernel that lasts > 10 ms on
big = [e for e in ev if "copyBuffer" in e[2] and e[1] - e[0] > 10_000_000]
if not big:
    print("no big copies"); raise SystemExit
t0 = big[0][0] - 100_000_000
sec = [e for e in ev if e[0] >= t0]
mainq = max(set(e[3] for e in sec if "copy" not in e[3].lower() and "copyBuffer" not in e[2]), key=lambda q: sum(e[1] - e[0] for e in sec if e[3] == q))
prev_end = None
for e in sec:
    if e[3] != mainq:
        continue
    if prev_end is not None and e[0] - prev_end > 3_000_000:
        during = [(x[2], round((min(x[1], e[0]) - max(x[0], prev_end)) / 1e6, 1), x[3]) for x in sec if x[3] != mainq and x[1] > prev_end and x[0] < e[0]]
        print(f"  main queue {mainq} idle {(e[0] - prev_end) / 1e6:7.1f} ms before {e[2][:40]}; meanwhile: {during[:4]}")
    prev_end = e[1] if prev_end is None else max(prev_end, e[1])
PY
  rm -rf $OUT/t$i
done
