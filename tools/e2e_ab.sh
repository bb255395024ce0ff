#!/bin/bash
# host-pointer path (r2s_sdf) under several environment settings: tools/e2e_ab.sh "A=1" "B=2" ...
for v in "$@"; do
  env $v R2S_HOST_TIMING=1 python bench.py --no-build --no-cpu-baseline --steps 5 > gpurun_out/b.json 2> gpurun_out/b.err
  echo "== $v"; grep -a "r2s host" gpurun_out/b.err | tail -2 | cut -c1-200
  python -c "
import json; d=json.load(open('gpurun_out/b.json'))['e2e']; print('pinned %.2f ms pageable %.2f ms' % (d['pinned']['ms_per_call'], d['pageable']['ms_per_call']), d['pinned']['equals_device_path'], d['pageable']['equals_device_path'])"
done
