# Two ranks on ONE GPU with the gloo backend: bookkeeping rehearsal of every --gather mode of bench.py at N = 2 (the data of the
# collectives is staged through the host by gloo; `direct` moves rows by peer-mapped stores).  Usage (GPU box): bash tools/rehearse_n2_one_gpu.sh <tag>
TAG=${1:-n2}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for G in auto dgrad expand direct mesh coef; do
  # round 5: started as the driver starts it -- `python3 bench.py --gpus 2 ...`, NO launcher: bench.py starts its own ranks
  timeout -k 10 400 python3 bench.py --gpus 2 --backend gloo --gather $G \
      --clips-per-gpu 4 --steps 2 --warmup 1 > $OUT/n2_$G.json 2> $OUT/n2_$G.err || { echo "FAILED $G"; tail -5 $OUT/n2_$G.err; }
  python3 -c "
import json; d=json.loads(open('$OUT/n2_$G.json').read().strip().splitlines()[-1]); print('$G', d['value'], d['ms_per_step'], d['config']['gather'], d['config']['gather_checksum_ok'], d['config'].get('gather_auto_ms_per_step'), 'world seen', d['config']['world_size_seen'], 'launcher', d['config']['launcher'], 'devices', [(r['rank'], r['pci_bus_id']) for r in d['config']['devices']])" || true
done
