#!/bin/bash
# SQ instruction counters of the raw-SDF kernels on config 5 (TET4, 1024^3):  tools/quick_sq_tet.sh <tag>
set -e
TAG=${1:-qt}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG && mkdir -p $OUT/prof_$TAG
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/prof_$TAG/sq1 -- python3 $ROOT/tools/profile_modes.py --tet --modes sdf --reps 2 > /dev/null
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/prof_$TAG/sq2 -- python3 $ROOT/tools/profile_modes.py --tet --modes sdf --reps 2 > /dev/null
python3 $ROOT/tools/pmc_summary.py $OUT/prof_$TAG/sq1 $OUT/prof_$TAG/sq2 --out $OUT/${TAG}_valu_counters.json > $OUT/${TAG}_valu_counters.txt
rm -rf $OUT/prof_$TAG
grep -A18 "^void iso_project_kernel\|^iso_project_kernel\|sdf_tiles_kernel<r2s::TetRec" $OUT/${TAG}_valu_counters.txt | head -80
