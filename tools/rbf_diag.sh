#!/bin/bash
# matvec time of the timing-only variants of the row-walk kernel:  tools/rbf_diag.sh 2 12 22 32 42
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for d in "$@"; do
  R2S_RBF_WALK_DIAG=$d bash $ROOT/tools/rbf_prof.sh 2>/dev/null | grep "rbf_walk_kernel<2, 7, 0" | sed "s/^/diag $d: /"
done
