#!/bin/bash
# kernel-trace stats of the RBF stage (tools/rbf_bench.py --interp) -> gpurun_out/rbf_stats.csv
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/rbfprof
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rbfprof -- python3 $ROOT/tools/rbf_bench.py --interp --reps 2 "$@" > $OUT/rbf_bench_trace.json
cp $(ls $OUT/rbfprof/*/*kernel_stats.csv | head -1) $OUT/rbf_stats.csv
cp $(ls $OUT/rbfprof/*/*kernel_trace.csv | head -1) $OUT/rbf_trace.csv
rm -rf $OUT/rbfprof
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/rbf_stats.csv")):
    if float(r["TotalDurationNs"]) > 2e6:
        print(f'{r["Name"][:60]:60s} {int(r["Calls"]):4d} {float(r["AverageNs"])/1e6:8.3f} ms  total {float(r["TotalDurationNs"])/1e6:8.2f}')
PY
