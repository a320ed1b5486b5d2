# Which engine moves the rows of the PCIe-inclusive twin?  The runtime's blit kernel (__amd_rocclr_copyBuffer: on the CUs, beside the
# persistent kernels) is what a trace shows; this tries the runtime's switches for the copy path.  Usage (GPU box): bash tools/ab_copy_engine.sh
cd $GRAFT_REPO_ROOT
B="--steps 4 --warmup 1 --no-cpu-baseline --no-mixed-precision --no-surface --no-column-sharing"
for E in "X=0" "GPU_FORCE_BLIT_COPY_SIZE=0" "DEBUG_CLR_LIMIT_BLIT_WG=8" "DEBUG_CLR_LIMIT_BLIT_WG=64" "HSA_ENABLE_SDMA=0" "X=0"; do
  env $E timeout -k 10 300 python3 bench.py $B > /tmp/o.json 2>/tmp/o.err || { echo "$E FAILED"; tail -2 /tmp/o.err; continue; }
  python3 -c "
import json; d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); w=d['with_h2d_d2h']
print('$E: device-resident', d['ms_per_step'], 'with_h2d_d2h', w['ms_per_step'], w['value'], 'd2h alone GB/s', w.get('d2h_alone_gbps'))"
done
