#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace (+ --memory-copy-trace) run: for every dispatch / copy NOT on the main compute queue, how much
of its duration ran while a kernel of the main queue was executing.  Answers "does a second-stream kernel (RCCL's all-gather,
a blit copy) get to run under the persistent one-workgroup-per-CU kernels?" (VERDICT r2 item 3c).

Usage: python profiles/overlap.py <kernel_trace.csv> [<memory_copy_trace.csv>]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    byq = collections.defaultdict(list)
    for r in rows:
        byq[(r.get("Queue_Id"), r.get("Stream_Id", ""))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    main_q = max(byq, key=lambda q: sum(e - s for s, e, _ in byq[q]))
    main_iv = sorted((s, e) for s, e, _ in byq[main_q])
    print(f"main queue {main_q}: {len(main_iv)} dispatches, busy {sum(e - s for s, e in main_iv) / 1e6:.1f} ms")

    def covered(s, e):
        tot = 0
        for a, b in main_iv:
            if b <= s:
                continue
            if a >= e:
                break
            tot += min(b, e) - max(a, s)
        return tot

    agg = collections.defaultdict(lambda: [0, 0, 0])
    for q, lst in byq.items():
        if q == main_q:
            continue
        for s, e, name in lst:
            k = (str(q), name.split("(")[0][:70])
            agg[k][0] += 1; agg[k][1] += e - s; agg[k][2] += covered(s, e)
    if len(sys.argv) > 2:
        for r in csv.DictReader(open(sys.argv[2])):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            k = ("copy", r.get("Direction", r.get("Name", "?")))
            agg[k][0] += 1; agg[k][1] += e - s; agg[k][2] += covered(s, e)
    print(f"{'queue':>22s} {'kernel / copy':70s} {'calls':>6s} {'total ms':>10s} {'under main-queue kernels':>26s}")
    for (q, name), (n, dur, cov) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if dur < 1e5:
            continue
        print(f"{q:>22s} {name:70s} {n:6d} {dur / 1e6:10.2f} {100.0 * cov / dur:25.1f}%")


if __name__ == "__main__":
    main()
