# The secondary bench lines of a round, in ONE gpurun call.  Usage (GPU box): bash tools/collect_extras.sh <tag>
# Output: gpurun_out/<tag>/ ; install with tools/install_profiles.sh <tag> <rNN>
TAG=${1:-extras}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
Q="--no-cpu-baseline --no-mixed-precision --no-host-io --no-surface"
# the headline line again, now that profiles/<round>_pmc of THIS tree is installed (roofline.traffic, frontend / attention counter figures live)
python3 bench.py --steps 5 --warmup 1 > $OUT/bench_final.json 2> $OUT/err.txt; echo final-done
python3 bench.py --steps 20 --warmup 3 > $OUT/bench_steps20.json 2>> $OUT/err.txt; echo steps20-done
python3 bench.py --sample-rate 8000 $Q > $OUT/bench_8khz.json 2>> $OUT/err.txt; echo 8khz-done
python3 bench.py --head offsets --ragged-seconds 3,6 --sample-rate 8000 --clips-per-gpu 80 $Q > $OUT/bench_config5.json 2>> $OUT/err.txt; echo config5-done
python3 bench.py --mesh-stage $Q --no-column-sharing > $OUT/bench_mesh_stage.json 2>> $OUT/err.txt; echo mesh-done
# BASELINE configs[0] regime: one clip per step (2 s and 10 s), device-resident like the headline, per-stage times in the line
python3 bench.py --steps 20 --warmup 3 --clips-per-gpu 1 --seconds 2 $Q > $OUT/bench_1x2s.json 2>> $OUT/err.txt; echo 1x2s-done
python3 bench.py --steps 20 --warmup 3 --clips-per-gpu 1 --seconds 10 $Q > $OUT/bench_1x10s.json 2>> $OUT/err.txt; echo 1x10s-done
# one-GPU rehearsal of the N > 1 exchange through RCCL (world size 1), with the hardware-queue settings bench.py now makes
for G in dgrad expand; do
  python3 bench.py --steps 5 --warmup 2 $Q --no-column-sharing --force-gather --gather $G --backend nccl > $OUT/rehearsal_nccl_w1_$G.json 2>> $OUT/err.txt
done
python3 bench.py --steps 5 --warmup 2 $Q --no-column-sharing > $OUT/rehearsal_nccl_w1_plain.json 2>> $OUT/err.txt
python3 bench.py --steps 5 --warmup 2 $Q --no-column-sharing --force-gather --gather dgrad --backend nccl --reserve-cus 16 > $OUT/rehearsal_nccl_w1_dgrad_reserve16.json 2>> $OUT/err.txt
echo rehearsal-done
python3 tools/time_split_lstm.py > $OUT/time_lstm_split.txt 2>> $OUT/err.txt
# round 4: front end same-process A/B (column FFT, numbering order, gather order) and the stress run of the cooperating-workgroup time LSTM
python3 tools/time_frontend.py 20 mel_fft_radix4=0,1 frontend_t_major=0,1 gather_plain_order=0,1 > $OUT/time_frontend.txt 2>> $OUT/err.txt; echo frontend-ab-done
python3 tools/stress_split_lstm.py 3000 > $OUT/stress_split_lstm.txt 2>> $OUT/err.txt; echo stress-done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{os.path.basename(f):40s} {d['value']:10.1f} frames/s {d['ms_per_step']:9.3f} ms/step gather={d['config'].get('gather')} checksum={d['config'].get('gather_checksum_ok')}")
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $OUT/err.txt
