#!/bin/bash
# usage: tools/ab.sh "<opt list A>" "<opt list B>" ...   (each arg = extra bench.py flags); prints value + stage times
mkdir -p gpurun_out
i=0
for args in "$@"; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err || { echo "FAIL: $args"; tail -3 gpurun_out/ab_$i.err; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$i.json"))
print("[$args]", d["value"], "lstm_frac", d["roofline"]["frac"], {k: round(v,2) for k,v in d["stage_ms_per_step"].items()})
PY
  i=$((i+1))
done
