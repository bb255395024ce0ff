#!/bin/bash
# SQ / TA / TCP counters of the RBF stage kernels (separate --pmc passes):  tools/rbf_pmc.sh <tag> [rbf_bench args]
set -e
TAG=${1:-rbf}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_$TAG && mkdir -p $OUT/pmc_$TAG
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU"
P3="TA_TA_BUSY TA_BUFFER_TOTAL_CYCLES"
P4="TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ"
P5="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_INST_LEVEL_VMEM"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_$TAG/p$i -- python3 $ROOT/tools/rbf_bench.py --interp --reps 1 "$@" > /dev/null
done
python3 $ROOT/tools/pmc_summary.py $OUT/pmc_$TAG/p1 $OUT/pmc_$TAG/p2 $OUT/pmc_$TAG/p3 $OUT/pmc_$TAG/p4 $OUT/pmc_$TAG/p5 --filter rbf_walk --out $OUT/${TAG}_pmc.json > $OUT/${TAG}_pmc.txt
rm -rf $OUT/pmc_$TAG
cat $OUT/${TAG}_pmc.txt
