#!/bin/bash
# A/B timing helpers (run on the GPU box from the repo root, after a build).  One script, four modes:
#   tools/ab.sh env   VAR [bench args]            bench line with VAR=0 / VAR=1, twice
#   tools/ab.sh vals  VAR "v1 v2 ..." [bench args] bench line under several values of one variable, twice
#   tools/ab.sh grid  "A=1 B=2" "A=0 B=2" ...      bench line under several settings (BENCH_ARGS for other workloads), twice
#   tools/ab.sh libs  diag/a.so diag/b.so base     bench line + one rank's share of 8 under several builds of the library
#   tools/ab.sh e2e   "A=1" "B=2" ...              host-pointer path (r2s_sdf) with its host phases
#   tools/ab.sh rbf   VAR v1 v2 ... [-- rbf_bench args]   kernel times of the RBF stage (tools/rbf_prof.sh) per value
set -u
mkdir -p gpurun_out
MODE=${1:?mode}; shift
bench_line() {   # $1 label, $2 "env settings", rest: bench args
  local label=$1 envs=$2; shift 2
  env $envs timeout -k 10 200 python bench.py --no-build --steps 30 --warmup 4 --no-cpu-baseline --no-e2e "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err \
    || { echo "$label failed"; tail -5 gpurun_out/ab.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/ab.json")); print("$label", round(d["ms_per_step"],3), {k: round(x,3) for k,x in d["stages_ms"].items()})
PY
}
case $MODE in
  env)  VAR=$1; shift; for v in 0 1 0 1; do bench_line "$VAR=$v" "$VAR=$v" "$@"; done ;;
  vals) VAR=$1; VALS=$2; shift 2; for rep in 1 2; do for v in $VALS; do bench_line "$VAR=$v" "$VAR=$v" "$@"; done; done ;;
  grid) for rep in 1 2; do for v in "$@"; do bench_line "$v" "$v" ${BENCH_ARGS:-}; done; done ;;
  libs) for rep in 1 2; do for v in "$@"; do
          lib=$v; [ "$v" = base ] && lib=rho2sdf.jl_amd/librho2sdf_hip.so
          bench_line "$v" "R2S_LIB_OVERRIDE=$lib" ${BENCH_ARGS:-}
          R2S_LIB_OVERRIDE=$lib WORLDS=8 timeout -k 10 200 python tools/rank_share.py 2>> gpurun_out/ab.err | tail -1 | cut -c1-220
        done; done ;;
  e2e)  for v in "$@"; do
          env $v R2S_HOST_TIMING=1 timeout -k 10 300 python bench.py --no-build --no-cpu-baseline --steps 5 > gpurun_out/b.json 2> gpurun_out/b.err \
            || { echo "$v failed"; tail -5 gpurun_out/b.err; exit 1; }
          echo "== $v"; grep -a "r2s host" gpurun_out/b.err | tail -2 | cut -c1-200
          python -c "
import json; d=json.load(open('gpurun_out/b.json'))['e2e']
for k in ('pageable', 'pinned'): print(k, '%.2f ms best, %.2f median' % (d[k]['ms_per_call'], d[k]['ms_per_call_median']), d[k]['host_phases_ms_median'])
print(d['rho2sdf_default_options']['stages_ms'])"
        done ;;
  rbf)  VAR=$1; shift; VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "${1:-}" == "--" ] && shift
        for v in "${VALS[@]}"; do env $VAR=$v bash tools/rbf_prof.sh "$@" 2>/dev/null | grep "rbf_walk_kernel\|cg_update\|volume_rowwave" | sed "s/^/$VAR=$v: /"; done ;;
  *) echo "unknown mode $MODE"; exit 2 ;;
esac
