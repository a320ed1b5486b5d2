# Repeats the PCIe-inclusive twin in fresh processes: does the probed copy stream (sdfa_amd/streams.py) overlap every time?
# Usage (GPU box): bash tools/ab_hostio.sh [runs]
cd $GRAFT_REPO_ROOT
B="--steps 4 --warmup 1 --no-cpu-baseline --no-mixed-precision --no-surface --no-column-sharing"
N=${1:-8}
for i in $(seq 1 $N); do
  Q=$(( (i % 2) * 12 + 4 ))
  GPU_MAX_HW_QUEUES=$Q python3 bench.py $B > /tmp/o.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); p=d['with_h2d_d2h']['copy_stream_overlaps_kernels']
print('run $i queues $Q: device-resident', d['ms_per_step'], 'with_h2d_d2h', d['with_h2d_d2h']['ms_per_step'], d['with_h2d_d2h']['value'], 'probes', [(q['priority'], q['overlaps'], q['copy_end_ms']) for q in p['probes']])"
done
