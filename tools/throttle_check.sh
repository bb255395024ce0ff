#!/bin/bash
# does the cgroup CPU quota of the box throttle the host side of the e2e legs?  (cpu.stat before / after a bench run under
# several thread settings; run on the GPU box from the repo root)
thr() { grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; }
for cfg in "R2S_HOST_THREADS=16" "R2S_HOST_THREADS=16 OMP_NUM_THREADS=1" "R2S_HOST_THREADS=12" "R2S_HOST_THREADS=16 OMP_NUM_THREADS=4" "R2S_HOST_THREADS=16 HIP_LAUNCH_BLOCKING=0 GPU_MAX_HW_QUEUES=4"; do
  echo "== $cfg"; thr
  env $cfg timeout -k 10 300 python bench.py --no-build --no-cpu-baseline --steps 5 > gpurun_out/b.json 2> gpurun_out/b.err || { echo failed; tail -3 gpurun_out/b.err; exit 1; }
  thr
  python -c "
import json; d=json.load(open('gpurun_out/b.json'))['e2e']
for k in ('pageable', 'pinned'): print(k, '%.2f ms best, %.2f median' % (d[k]['ms_per_call'], d[k]['ms_per_call_median']), {a: b for a, b in d[k]['host_phases_ms_median'].items() if a in ('wait_fill', 'scatter')})
print(d['rho2sdf_default_options']['stages_ms'])"
done
