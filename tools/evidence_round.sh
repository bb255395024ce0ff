#!/bin/bash
# the GPU evidence of a round in one gpurun call (after a build and tools/build_diag.sh stats -DR2S_ISO_STATS,
# strag -DR2S_STRAG_DIAG=1, strag2 -DR2S_STRAG_DIAG=2):
#   tools/evidence_round.sh r03   -> gpurun_out/<tag>_*; copy what is to be judged into profiles/
set -e
TAG=${1:-r03}
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/${TAG}_gpu_tests.log 2>&1; tail -2 $O/${TAG}_gpu_tests.log
python bench.py > $O/${TAG}_bench_driver_style.json 2> $O/bench.err; echo "bench done"
python bench.py --no-build --workload chapadlo256 --no-e2e > $O/${TAG}_bench_chapadlo256.json 2>> $O/bench.err; echo "chapadlo256 done"
python bench.py --no-build --workload tet5 --no-e2e > $O/${TAG}_bench_tet5.json 2>> $O/bench.err; echo "tet5 done"
R2S_LIB_OVERRIDE=diag/stats.so python tools/iso_phase_stats.py > $O/${TAG}_iso_phase_stats.txt 2> $O/diag.err; echo "phase stats done"
R2S_LIB_OVERRIDE=diag/strag.so python tools/strag_diag.py > $O/${TAG}_straggler_waves.txt 2>> $O/diag.err
R2S_LIB_OVERRIDE=diag/strag2.so python tools/strag_diag.py hist >> $O/${TAG}_straggler_waves.txt 2>> $O/diag.err
R2S_LIB_OVERRIDE=diag/strag.so python tools/strag_diag.py chapadlo256 > $O/${TAG}_straggler_waves_chapadlo256.txt 2>> $O/diag.err; echo "strag diag done"
bash tools/profile_round.sh $TAG
