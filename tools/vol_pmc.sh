#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_vol && mkdir -p $OUT/pmc_vol
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_vol/p$i -- python3 $ROOT/tools/rbf_bench.py --interp --reps 1 > /dev/null
done
python3 $ROOT/tools/pmc_summary.py $OUT/pmc_vol/p1 $OUT/pmc_vol/p2 --filter volume_rowwave
rm -rf $OUT/pmc_vol
