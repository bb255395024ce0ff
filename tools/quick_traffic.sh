#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel of one north-star step (two --pmc passes):  tools/quick_traffic.sh <tag>
set -e
TAG=${1:-qtr}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG && mkdir -p $OUT/prof_$TAG
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_$TAG/fetch -- python3 $ROOT/tools/profile_modes.py --modes sdf --reps 2 > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_$TAG/write -- python3 $ROOT/tools/profile_modes.py --modes sdf --reps 2 > /dev/null
python3 $ROOT/tools/collect_traffic.py $OUT/prof_$TAG/fetch $OUT/prof_$TAG/write $OUT/${TAG}_traffic.json
rm -rf $OUT/prof_$TAG
python3 - <<PY
import json
d=json.load(open("$OUT/${TAG}_traffic.json")); print("step bytes", d["step_bytes"])
for k,v in d["kernels"].items():
    if k.startswith("sign_project") or k.startswith("sdf_tiles"): print(k[:60], v)
PY
