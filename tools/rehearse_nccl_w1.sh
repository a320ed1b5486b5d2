# One-GPU rehearsal of the N > 1 exchange through RCCL (world size 1): step time with and without the concurrent per-chunk
# all-gather stream, and with CUs reserved for it.  Usage (GPU box): bash tools/rehearse_nccl_w1.sh <tag>
set -e
TAG=${1:-nccl_w1}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
B="--steps 5 --warmup 2 --no-cpu-baseline --no-mixed-precision --no-column-sharing --no-host-io --no-surface"
python3 $R/bench.py $B > $OUT/plain.json 2> $OUT/err.txt
for G in dgrad expand; do
  python3 $R/bench.py $B --force-gather --gather $G --backend nccl > $OUT/w1_$G.json 2>> $OUT/err.txt
done
for K in 8 16 32; do
  python3 $R/bench.py $B --force-gather --gather dgrad --backend nccl --reserve-cus $K > $OUT/w1_dgrad_reserve$K.json 2>> $OUT/err.txt
  python3 $R/bench.py $B --reserve-cus $K > $OUT/plain_reserve$K.json 2>> $OUT/err.txt
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{os.path.basename(f):32s} {d['value']:10.1f} frames/s  {d['ms_per_step']:8.2f} ms/step  gather={d['config']['gather']} reserved={d['config']['reserved_cus']} checksum={d['config']['gather_checksum_ok']} peak_mem={d.get('peak_device_memory_gb')}")
    except Exception as e:
        print(f, "ERR", e)
PY
