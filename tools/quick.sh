#!/bin/bash
# quick looks between kernel edits (GPU box, repo root, after a build; every rocprofv3 run under `timeout`):
#   tools/quick.sh stats   <tag> [bench args]        kernel-trace --stats of the bench command
#   tools/quick.sh cmd     <tag> <script> [args]     kernel-trace --stats of another python tool
#   tools/quick.sh sq      <tag> [--tet]             SQ instruction counters of the raw-SDF kernels (two --pmc passes)
#   tools/quick.sh traffic <tag>                     FETCH_SIZE / WRITE_SIZE per kernel of one north-star step
#   tools/quick.sh trace   <tag> [--tet]             kernel timeline of one step, window around the projection kernel
set -e
MODE=${1:?mode}; TAG=${2:-q}; shift; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof_$TAG
rm -rf $P && mkdir -p $P
RP="timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv"
case $MODE in
  stats)
    $RP --stats -d $P/stats -- python3 $ROOT/bench.py --no-build --steps 10 --warmup 2 --no-cpu-baseline --no-e2e "$@" > $OUT/${TAG}_bench_under_trace.json
    cp $(ls $P/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
    head -25 $OUT/${TAG}_kernel_stats.csv | cut -c1-150 ;;
  cmd)
    SCRIPT=$ROOT/$1; shift
    $RP --stats -d $P/stats -- python3 $SCRIPT "$@" > $OUT/${TAG}_out.txt
    cp $(ls $P/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv ;;
  sq)
    $RP --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS -d $P/sq1 -- python3 $ROOT/tools/profile_modes.py "$@" --modes sdf --reps 2 > /dev/null
    $RP --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $P/sq2 -- python3 $ROOT/tools/profile_modes.py "$@" --modes sdf --reps 2 > /dev/null
    python3 $ROOT/tools/pmc_summary.py $P/sq1 $P/sq2 --out $OUT/${TAG}_valu_counters.json > $OUT/${TAG}_valu_counters.txt
    grep -A18 "^iso_project\|^iso_straggler_kernel\|^sign_project_kernel\|sdf_tiles_kernel<r2s::TetRec" $OUT/${TAG}_valu_counters.txt | head -80 ;;
  traffic)
    $RP --pmc FETCH_SIZE -d $P/fetch -- python3 $ROOT/tools/profile_modes.py --modes sdf --reps 2 > /dev/null
    $RP --pmc WRITE_SIZE -d $P/write -- python3 $ROOT/tools/profile_modes.py --modes sdf --reps 2 > /dev/null
    python3 $ROOT/tools/collect_traffic.py $P/fetch $P/write $OUT/${TAG}_traffic.json
    python3 -c "
import json
d=json.load(open('$OUT/${TAG}_traffic.json')); print('step bytes', d['step_bytes'])
for k,v in d['kernels'].items():
    if k.startswith('sign_project') or k.startswith('sdf_tiles') or k.startswith('iso_project'): print(k[:60], v)" ;;
  trace)
    $RP -d $P/tr -- python3 $ROOT/tools/profile_modes.py "$@" --modes sdf --reps 5 > /dev/null
    cp $(ls $P/tr/*/*kernel_trace.csv | head -1) $OUT/${TAG}_trace.csv
    if [ "${1:-}" = "--tet" ]; then python3 $ROOT/tools/trace_window.py $OUT/${TAG}_trace.csv "void iso_project_kernel" 30 45
    else python3 $ROOT/tools/trace_window.py $OUT/${TAG}_trace.csv "void iso_project_hex_pl" 44 10; fi ;;
  *) echo "unknown mode $MODE"; exit 2 ;;
esac
rm -rf $P
