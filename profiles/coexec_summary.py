#!/usr/bin/env python3
"""Per-kernel table of the SQ co-execution counters collected by tools/pmc_coexec.sh.

Usage: python profiles/coexec_summary.py <dir with <label>_p<N>.csv>
Every row is one (run label, kernel symbol, grid) with the counters summed over the kernel's dispatches of that run and
divided by the number of dispatches.  coexec % = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_BUSY_CYCLES-normalised columns are printed
raw as well: the counters are per-SE/XCD aggregates whose normalisation differs, so compare builds, not columns."""
import collections
import csv
import glob
import os
import re
import sys


def main():
    d = sys.argv[1]
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sorted(glob.glob(os.path.join(d, "*_p*.csv"))):
        label = re.sub(r"_p\d+\.csv$", "", os.path.basename(path))
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if "at::" in name or "elementwise" in name:
                continue
            rows[(label, name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = ["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES",
             "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU_MFMA_MOPS", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"]
    print("label | kernel | grid | calls | " + " | ".join(n.replace("SQ_", "") for n in names) +
          " | coexec/mfma_busy | wait_any/wave_cycles | wait_inst/wave_cycles | active/wave_cycles")
    for key in sorted(rows, key=lambda k: (k[1], k[2], k[0])):
        c = rows[key]
        if max((sum(v) / len(v) for k, v in c.items() if k == "SQ_INSTS_MFMA"), default=0) < 1e5:
            continue
        avg = {n: (sum(c[n]) / len(c[n]) if c.get(n) else float("nan")) for n in names}
        calls = max(len(v) for v in c.values())

        def ratio(a, b):
            return avg[a] / avg[b] if avg[b] == avg[b] and avg[b] else float("nan")
        print(f"{key[0]} | {key[1][:60]} | {key[2]} | {calls} | " + " | ".join(f"{avg[n]:.4g}" for n in names) +
              f" | {ratio('SQ_VALU_MFMA_COEXEC_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES'):.4f} | {ratio('SQ_WAIT_ANY', 'SQ_WAVE_CYCLES'):.4f}"
              f" | {ratio('SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES'):.4f} | {ratio('SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES'):.4f}")


if __name__ == "__main__":
    main()
