#!/bin/bash
# NS step (bench.py) and one rank's share of 8 (tools/rank_share.py) under several builds of the library:
#   tools/ab_libs.sh diag/a.so diag/b.so ...   ("base" = the built library)
for rep in 1 2; do
for v in "$@"; do
  lib=$v; [ "$v" = base ] && lib=rho2sdf.jl_amd/librho2sdf_hip.so
  R2S_LIB_OVERRIDE=$lib timeout -k 10 200 python bench.py --no-build --steps 30 --warmup 4 --no-cpu-baseline --no-e2e $BENCH_ARGS > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$v failed"; tail -5 gpurun_out/ab.err; exit 1; }
  R2S_LIB_OVERRIDE=$lib WORLDS=8 timeout -k 10 200 python tools/rank_share.py > gpurun_out/ab_rs.json 2>> gpurun_out/ab.err
  python - <<PY
import json; d=json.load(open("gpurun_out/ab.json")); r=json.loads(open("gpurun_out/ab_rs.json").read().strip().splitlines()[-1])
print("%-14s NS %.3f (fast %.3f main %.3f)   rank of 8: %.3f (fast %.3f main %.3f)" % ("$v", d["ms_per_step"], d["stages_ms"]["ms_iso_fast"], d["stages_ms"]["ms_main"], r["ms_run"], r["stages"]["ms_iso_fast"], r["stages"]["ms_main"]))
PY
done
done
