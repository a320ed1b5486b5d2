#!/bin/bash
# usage: tools/ab2.sh "<ENV=.. list|-> :: <bench flags>" ...   each arg: optional env assignments, then '::', then bench.py flags
mkdir -p gpurun_out
i=0
for spec in "$@"; do
  envs="${spec%%::*}"; args="${spec#*::}"
  [ "$envs" = "- " ] && envs=""
  env $envs timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-column-sharing --no-mixed-precision $args > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err || { echo "FAIL: $spec"; tail -3 gpurun_out/ab_$i.err; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$i.json"))
print("[$spec]", d["value"], "lstm_frac", d["roofline"]["frac"], {k: round(v,2) for k,v in d["stage_ms_per_step"].items()})
PY
  i=$((i+1))
done
