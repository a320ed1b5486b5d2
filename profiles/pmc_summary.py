#!/usr/bin/env python3
"""Per-kernel, per-launch-shape table from the three separate rocprofv3 --pmc passes (MfmaUtil, FETCH_SIZE, WRITE_SIZE).

Usage: python profiles/pmc_summary.py <dir with MfmaUtil_/FETCH_SIZE_/WRITE_SIZE_counter_collection.csv> [--traffic-json out.json]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB on gfx950 (MI355X_MICROARCH.md, HBM section); the table shows
them per launch.  --traffic-json writes the freq_lstm_kernel<false,..> figures of the 8192-frame launch, which
bench.py scales to its frames per launch for `roofline.traffic`."""
import collections
import csv
import json
import os
import sys


def load(path):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out[(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return out


d = sys.argv[1]
M, F, W = (load(os.path.join(d, f"{c}_counter_collection.csv")) for c in ("MfmaUtil", "FETCH_SIZE", "WRITE_SIZE"))
print(f"{'kernel':46s} {'grid':>10s} {'calls':>5s} {'MfmaUtil %':>10s} {'FETCH KiB':>12s} {'WRITE KiB':>12s}")
for k in sorted(M, key=lambda k: -(sum(F.get(k, [0])) + sum(W.get(k, [0])))):
    f, w, m = F.get(k, [0]), W.get(k, [0]), M[k]
    if sum(f) + sum(w) < 1e5 or "at::" in k[0]:
        continue
    print(f"{k[0][:46]:46s} {k[1]:10d} {len(m):5d} {sum(m) / len(m):10.1f} {sum(f) / len(f):12.0f} {sum(w) / len(w):12.0f}")
if "--traffic-json" in sys.argv:
    key = next(k for k in M if k[0].startswith("freq_lstm_kernel<false") and k[1] == 8192 * 64 // 64 * 2 * 256)
    fk, wk = sum(F[key]) / len(F[key]), sum(W[key]) / len(W[key])
    alg = 8192 * (512 * 1024 + 2 * 1024 * 1024)
    json.dump({"kernel": key[0], "frames": 8192, "fetch_kib": round(fk), "write_kib": round(wk),
               "source": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/r01_pmc/)",
               "algorithmic_bytes": alg,
               "note": f"algorithmic = conv3 activations read once (512 KiB/frame) + hidden states written once (2 MiB/frame) = "
                       f"{alg / 1e9:.2f} GB; measured read {fk * 1024 / 1e9:.2f} GB + write {wk * 1024 / 1e9:.2f} GB "
                       f"(the 0.77 MB of weights are served by L2)"},
              open(sys.argv[sys.argv.index("--traffic-json") + 1], "w"), indent=1)
