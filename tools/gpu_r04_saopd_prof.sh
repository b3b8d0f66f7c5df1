#!/bin/bash
# round-4: phase timers of saopd_wave_kernel (MP_PROFILE build; planner 0 prints clock64 ticks per phase), first and following plans
cd /root/repo
mkdir -p gpurun_out/r04
export MP_PROFILE=1
for d in ${DICTS:-0 1}; do
  MP_SAOPD_DICT=$d timeout 300 python bench.py --workload saopd --no-cpu-baseline --no-parity-sample --steps 1 --warmup 0 > gpurun_out/r04/saopd_prof$d.txt 2> gpurun_out/r04/saopd_prof$d.err
  echo "== dict=$d"; grep "saopd prof" gpurun_out/r04/saopd_prof$d.txt | tail -12
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/r04/saopd_prof$d.txt') if l.startswith('{')][-1])
    print('kernel_ms', d['roofline'].get('kernel_ms'), d['config'].get('kernel_ms_first_and_following_plans'))
except Exception as e: print('ERR', e); print(open('gpurun_out/r04/saopd_prof$d.err').read()[-1500:])
PY
done
