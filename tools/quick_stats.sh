#!/bin/bash
# kernel-trace --stats of the bench command only (a quick look between kernel edits):  tools/quick_stats.sh <tag> [bench args]
set -e
TAG=${1:-q}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG && mkdir -p $OUT/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG/stats -- python3 $ROOT/bench.py --no-build --steps 10 --warmup 2 --no-cpu-baseline --no-e2e "$@" > $OUT/${TAG}_bench_under_trace.json
cp $(ls $OUT/prof_$TAG/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/prof_$TAG
head -25 $OUT/${TAG}_kernel_stats.csv | cut -c1-150
