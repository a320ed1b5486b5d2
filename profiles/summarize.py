#!/usr/bin/env python3
"""Per-kernel, per-launch-shape averages from a rocprofv3 --kernel-trace CSV (profiles/*_kernel_trace.csv).

rocprofv3's --stats averages every launch of a kernel symbol; bench.py launches the hot kernels in chunks of 8192,
8192 and 3968 frames (and once more on the 2 s parity sample), so the figure that corresponds to `roofline.launch_ms`
is the average over the three chunk launches of a step.  Round 2: the persistent forms of the frequency LSTM launch the
same grid whatever the chunk size, and `sdfa_model_autotune` adds 12 launches of 8192 frames (4 forms x 3) in front of the
first step -- the last block below lists the launches INSIDE steps (those between two `conv123_kernel` launches of the
headline pass) and their mean, which is what bench.py's `roofline.launch_ms` measures with HIP events.
Usage: python profiles/summarize.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    acc[(name, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
tot = collections.defaultdict(lambda: [0, 0.0])
print(f"{'kernel':48s} {'grid':>10s} {'calls':>6s} {'avg ms':>10s}")
for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) < 0.05:
        continue
    print(f"{name[:48]:48s} {grid:10d} {len(v):6d} {sum(v) / len(v):10.4f}")
    tot[name][0] += len(v); tot[name][1] += sum(v)
print()
for name, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:48]:48s} all launches: {n:5d} calls, avg {t / n:9.4f} ms, total {t:9.2f} ms")

# frequency-LSTM launches that belong to steps: every conv123_kernel launch (one per chunk) is followed by exactly one of them
seq = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
clean = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
in_step = collections.defaultdict(list)
want = False
for r in seq:
    n = clean(r)
    if n.startswith("conv123_kernel"):
        want = True
    elif want and n.startswith(("freq_lstm_v3_kernel", "freq_lstm_v2_kernel", "freq_lstm_kernel", "freq_lstm_bf16_kernel")):
        in_step[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        want = False
print()
for n, v in sorted(in_step.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n[:48]:48s} launches inside steps (autotune excluded): {len(v):4d} calls, avg {sum(v) / len(v):9.4f} ms")
