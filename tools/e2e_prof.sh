#!/bin/bash
# kernel-trace stats of whole rho2sdf() calls (tools/e2e_wall.py) -> gpurun_out/e2e_stats.csv
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/e2eprof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2eprof -- python3 $ROOT/tools/e2e_wall.py > $OUT/e2e_wall_trace.txt
cp $(ls $OUT/e2eprof/*/*kernel_stats.csv | head -1) $OUT/e2e_stats.csv
cp $(ls $OUT/e2eprof/*/*kernel_trace.csv | head -1) $OUT/e2e_trace.csv
rm -rf $OUT/e2eprof
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/e2e_stats.csv")):
    if float(r["TotalDurationNs"]) > 6e6:
        print(f'{r["Name"][:64]:64s} {int(r["Calls"]):5d} {float(r["AverageNs"])/1e6:8.3f} ms  per call {float(r["TotalDurationNs"])/6e6:8.2f}')
PY
