#!/bin/bash
# state-aware OPD: following plans with the grouped backup forced on / off
cd /root/repo
mkdir -p gpurun_out/r04
for pb in "" 1 0; do
  if [ -n "$pb" ]; then export MP_SAOPD_PAR_BACKUP=$pb; else unset MP_SAOPD_PAR_BACKUP; fi
  timeout 300 python bench.py --workload saopd --no-cpu-baseline --no-parity-sample --steps 3 --warmup 1 > gpurun_out/r04/saopd_pb.json 2> gpurun_out/r04/saopd_pb.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/r04/saopd_pb.json') if l.startswith('{')][-1])
    print('PAR_BACKUP=[$pb] kernel_ms %.3f first+following %s'%(d['roofline'].get('kernel_ms'), d['config'].get('kernel_ms_first_and_following_plans')))
except Exception as e: print('ERR', e); print(open('gpurun_out/r04/saopd_pb.err').read()[-1500:])
PY
done
