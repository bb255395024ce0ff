#!/bin/bash
# kernel timeline of one config-5 step (tools/profile_modes.py --tet --modes sdf) -> gpurun_out/tet_trace.csv + a window print
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tetprof
rocprofv3 --kernel-trace --output-format csv -d $OUT/tetprof -- python3 $ROOT/tools/profile_modes.py --tet --modes sdf --reps 4 "$@" > /dev/null
cp $(ls $OUT/tetprof/*/*kernel_trace.csv | head -1) $OUT/tet_trace.csv
rm -rf $OUT/tetprof
python3 $ROOT/tools/trace_window.py $OUT/tet_trace.csv "void iso_project_kernel" 30 45
