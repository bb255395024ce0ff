#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02
# -> gpurun_out/<tag>_kernel_stats.csv      kernel-trace --stats of `python3 bench.py` (same command as the bench line)
#    gpurun_out/<tag>_traffic.json          FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes)
#    gpurun_out/<tag>_valu_counters.json    SQ instruction / lane / FP64 counters per kernel (their own pass)
# The library must be built beforehand (nothing builds under the profiler: bench.py --no-build, tools use load_built).
set -e
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG && mkdir -p $OUT/prof_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG/stats -- python3 $ROOT/bench.py --no-build --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $OUT/${TAG}_bench_under_trace.json
cp $(ls $OUT/prof_$TAG/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_$TAG/fetch -- python3 $ROOT/tools/profile_modes.py --modes sdf --reps 2 > /dev/null
echo "fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_$TAG/write -- python3 $ROOT/tools/profile_modes.py --modes sdf --reps 2 > /dev/null
echo "write done"
python3 $ROOT/tools/collect_traffic.py $OUT/prof_$TAG/fetch $OUT/prof_$TAG/write $OUT/${TAG}_traffic.json
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/prof_$TAG/sq1 -- python3 $ROOT/tools/profile_modes.py --modes sdf --reps 2 > /dev/null
echo "sq1 done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/prof_$TAG/sq2 -- python3 $ROOT/tools/profile_modes.py --modes sdf --reps 2 > /dev/null
echo "sq2 done"
python3 $ROOT/tools/pmc_summary.py $OUT/prof_$TAG/sq1 $OUT/prof_$TAG/sq2 --out $OUT/${TAG}_valu_counters.json > $OUT/${TAG}_valu_counters.txt
# post-processing kernels (RBF smoothing with the CG, 512^3): kernel-trace stats + HBM traffic of the CG matvec
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG/rbf -- python3 $ROOT/tools/rbf_bench.py --interp --reps 2 > $OUT/${TAG}_rbf_bench.json
cp $(ls $OUT/prof_$TAG/rbf/*/*kernel_stats.csv | head -1) $OUT/${TAG}_rbf_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_$TAG/rbf_fetch -- python3 $ROOT/tools/rbf_bench.py --interp --reps 1 > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_$TAG/rbf_write -- python3 $ROOT/tools/rbf_bench.py --interp --reps 1 > /dev/null
python3 $ROOT/tools/collect_traffic.py $OUT/prof_$TAG/rbf_fetch $OUT/prof_$TAG/rbf_write $OUT/${TAG}_rbf_traffic.json
# the output field on a refined grid (rbf_grid = :fine, BASELINE config 3's shape: 257^3 -> 513^3)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG/rbf_fine -- python3 $ROOT/tools/rbf_bench.py --interp --grid 256 --smooth 2 --reps 2 > $OUT/${TAG}_rbf_fine_bench.json
cp $(ls $OUT/prof_$TAG/rbf_fine/*/*kernel_stats.csv | head -1) $OUT/${TAG}_rbf_fine_kernel_stats.csv
# ... and their SQ / TA / TCP counters (tools/rbf_pmc.sh: one --pmc pass per counter group)
bash $ROOT/tools/rbf_pmc.sh ${TAG}_rbf > /dev/null
# the whole rho2sdf() call: kernel stats per call
bash $ROOT/tools/e2e_prof.sh > $OUT/${TAG}_e2e_kernels.txt 2>/dev/null
cp $OUT/e2e_stats.csv $OUT/${TAG}_e2e_kernel_stats.csv
echo "rbf done"
rm -rf $OUT/prof_$TAG
echo "profile_round $TAG complete"
