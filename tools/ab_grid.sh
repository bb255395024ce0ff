#!/bin/bash
# bench line over a list of environment settings: tools/ab_grid.sh "A=1 B=2" "A=0 B=2" ... (each argument one setting;
# BENCH_ARGS="--workload chapadlo256" for another workload)
for rep in 1 2; do
for v in "$@"; do
  env $v timeout -k 10 200 python bench.py --no-build --steps 30 --warmup 4 --no-cpu-baseline --no-e2e $BENCH_ARGS > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$v failed"; tail -5 gpurun_out/ab.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/ab.json")); print("$v", round(d["ms_per_step"],3), {k: round(x,3) for k,x in d["stages_ms"].items()})
PY
done
done
