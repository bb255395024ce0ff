#!/bin/bash
# does the cgroup CPU quota of the box throttle the host side of the e2e legs?  (cpu.stat before / after a bench run under
# several settings; run on the GPU box from the repo root)    tools/throttle_check.sh "A=1 B=2" "A=0" ...
thr() { grep -E "usage_usec|nr_periods|nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; }
[ $# -eq 0 ] && set -- "R2S_HOST_THREADS=16" "R2S_HOST_THREADS=16 OMP_NUM_THREADS=1" "R2S_HOST_THREADS=12" "R2S_HOST_THREADS=16 OMP_NUM_THREADS=256"
for cfg in "$@"; do
  echo "== $cfg"; thr
  env $cfg timeout -k 10 300 python bench.py --no-build --no-cpu-baseline --steps 5 > gpurun_out/b.json 2> gpurun_out/b.err || { echo failed; tail -3 gpurun_out/b.err; exit 1; }
  thr
  python -c "
import json; d=json.load(open('gpurun_out/b.json'))['e2e']
for k in ('pageable', 'pinned'): print(k, '%.2f ms best, %.2f median' % (d[k]['ms_per_call'], d[k]['ms_per_call_median']), {a: b for a, b in d[k]['host_phases_ms_median'].items() if a in ('wait_fill', 'scatter')})
print(d['rho2sdf_default_options']['stages_ms'])"
done
