# Positive / negative control for SQ_VALU_MFMA_COEXEC_CYCLES: the register-only microbenchmarks in which one wave streams MFMAs
# while its SIMD partner runs vector-ALU work (tools/mfma_cowave.hip) or a wave issues vector instructions behind its own MFMAs
# (tools/mfma_shadow.hip), under the same counter pass as the product kernels.  Usage (GPU box): bash tools/pmc_coexec_control.sh <tag>
TAG=${1:-coexec_control}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for T in mfma_cowave mfma_shadow; do
  $R/tools/$T > $OUT/$T.stdout 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d $OUT/$T -o run -- $R/tools/$T > $OUT/$T.pmc.stdout 2> $OUT/$T.err
  F=$(find $OUT/$T -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && cp $F $OUT/$T.csv
  rm -rf $OUT/$T
done
python3 - <<PY
import csv, collections
for t in ("mfma_cowave", "mfma_shadow"):
    try:
        rows = list(csv.DictReader(open("$OUT/%s.csv" % t)))
    except Exception as e:
        print(t, "no csv", e); continue
    agg = collections.OrderedDict()
    for r in rows:
        k = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][-60:])
        agg.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    print("==", t, "(one line per dispatch, in launch order)")
    for (d, name), c in agg.items():
        print(f"{d:>4s} {name:60s} coexec={c.get('SQ_VALU_MFMA_COEXEC_CYCLES', float('nan')):.4g} insts_valu={c.get('SQ_INSTS_VALU', float('nan')):.4g} insts_mfma={c.get('SQ_INSTS_MFMA', float('nan')):.4g} busy={c.get('SQ_BUSY_CYCLES', float('nan')):.4g}")
PY
