#!/bin/bash
# kernel-trace --stats of an arbitrary python tool:  tools/quick_stats_cmd.sh <tag> <script> [args]   (env is inherited)
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG && mkdir -p $OUT/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG/stats -- python3 $SCRIPT "$@" > $OUT/${TAG}_out.txt
cp $(ls $OUT/prof_$TAG/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/prof_$TAG
