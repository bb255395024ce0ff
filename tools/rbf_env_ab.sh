#!/bin/bash
# kernel times of the RBF stage under settings of one environment variable:  tools/rbf_env_ab.sh VAR v1 v2 ... [-- rbf_bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
VAR=$1; shift
VALS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for v in "${VALS[@]}"; do
  env $VAR=$v bash $ROOT/tools/rbf_prof.sh "$@" 2>/dev/null | grep "rbf_walk_kernel\|cg_update" | sed "s/^/$VAR=$v: /"
done
