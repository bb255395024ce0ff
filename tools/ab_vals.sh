#!/bin/bash
# bench line under several values of one environment variable: tools/ab_vals.sh VAR "v1 v2 ..." [bench args]
VAR=$1; VALS=$2; shift; shift
for rep in 1 2; do
for v in $VALS; do
  env $VAR=$v timeout -k 10 200 python bench.py --no-build --steps 30 --warmup 4 --no-cpu-baseline --no-e2e "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$VAR=$v failed"; tail -5 gpurun_out/ab_$v.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/ab_$v.json")); print("$VAR=$v", round(d["ms_per_step"],3), {k: round(x,3) for k,x in d["stages_ms"].items()})
PY
done
done
