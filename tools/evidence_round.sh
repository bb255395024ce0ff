#!/bin/bash
# the GPU evidence of a round in one gpurun call (after a build and tools/build_diag.sh stats -DR2S_ISO_STATS,
# strag -DR2S_STRAG_DIAG=1, strag2 -DR2S_STRAG_DIAG=2 - the diagnostic parts are skipped when their builds are absent):
#   tools/evidence_round.sh r04   -> gpurun_out/<tag>_*; copy what is to be judged into profiles/
TAG=${1:-r04}
O=gpurun_out
mkdir -p $O
rc=0
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_gpu_tests.log 2>&1 || rc=$?
tail -2 $O/${TAG}_gpu_tests.log
[ $rc -eq 0 ] || { echo "GPU tests failed (rc $rc): no evidence collected"; exit $rc; }
step() {   # label, command...: stop at the first failure (after a GPU step timed out or was killed, no further GPU step)
  local label=$1; shift
  "$@" || { echo "$label failed (rc $?)"; tail -5 $O/bench.err 2>/dev/null; exit 1; }
  echo "$label done"
}
step bench        bash -c "timeout -k 10 600 python bench.py > $O/${TAG}_bench_driver_style.json 2> $O/bench.err"
step chapadlo256  bash -c "timeout -k 10 300 python bench.py --no-build --workload chapadlo256 --no-e2e > $O/${TAG}_bench_chapadlo256.json 2>> $O/bench.err"
step tet5         bash -c "timeout -k 10 300 python bench.py --no-build --workload tet5 --no-e2e > $O/${TAG}_bench_tet5.json 2>> $O/bench.err"
if [ -f diag/stats.so ]; then
  step phase_stats bash -c "R2S_LIB_OVERRIDE=diag/stats.so timeout -k 10 300 python tools/iso_phase_stats.py > $O/${TAG}_iso_phase_stats.txt 2> $O/diag.err"
fi
if [ -f diag/strag.so ] && [ -f diag/strag2.so ]; then
  step strag      bash -c "R2S_LIB_OVERRIDE=diag/strag.so timeout -k 10 300 python tools/strag_diag.py > $O/${TAG}_straggler_waves.txt 2>> $O/diag.err"
  step strag_hist bash -c "R2S_LIB_OVERRIDE=diag/strag2.so timeout -k 10 300 python tools/strag_diag.py hist >> $O/${TAG}_straggler_waves.txt 2>> $O/diag.err"
  step strag_c4   bash -c "R2S_LIB_OVERRIDE=diag/strag.so timeout -k 10 300 python tools/strag_diag.py chapadlo256 > $O/${TAG}_straggler_waves_chapadlo256.txt 2>> $O/diag.err"
fi
step profiles bash tools/profile_round.sh $TAG
