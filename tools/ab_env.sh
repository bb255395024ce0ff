#!/bin/bash
# A/B of one environment switch on the bench line: tools/ab_env.sh VAR "bench args"  -> ms_per_step with VAR=0 / VAR=1
VAR=$1; shift
for v in 0 1 0 1; do
  env $VAR=$v timeout -k 10 200 python bench.py --no-build --steps 30 --warmup 4 --no-cpu-baseline --no-e2e "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$VAR=$v failed"; tail -5 gpurun_out/ab_$v.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/ab_$v.json")); print("$VAR=$v", round(d["ms_per_step"],3), {k: round(x,3) for k,x in d["stages_ms"].items()})
PY
done
