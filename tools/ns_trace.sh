#!/bin/bash
# kernel timeline of one north-star step (tools/profile_modes.py --modes sdf) -> gpurun_out/ns_trace.csv + a window print
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/nsprof
rocprofv3 --kernel-trace --output-format csv -d $OUT/nsprof -- python3 $ROOT/tools/profile_modes.py --modes sdf --reps 5 "$@" > /dev/null
cp $(ls $OUT/nsprof/*/*kernel_trace.csv | head -1) $OUT/ns_trace.csv
rm -rf $OUT/nsprof
python3 $ROOT/tools/trace_window.py $OUT/ns_trace.csv "void iso_project_hex_pl" 44 10
