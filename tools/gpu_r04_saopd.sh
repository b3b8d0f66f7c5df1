#!/bin/bash
# round-4: state-aware OPD variants (dictionaries in LDS): parity tests + bench saopd, both forms
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q -k "state_aware or saopd or robust" > gpurun_out/r04/pytest_saopd.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04/pytest_saopd.log
tail -5 gpurun_out/r04/pytest_saopd.log
for d in 0 1; do
  MP_SAOPD_DICT=$d timeout 300 python bench.py --workload saopd --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r04/saopd_dict$d.json 2> gpurun_out/r04/saopd_dict$d.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/r04/saopd_dict$d.json') if l.startswith('{')][-1])
    print('dict=$d value %.4g ms %.3f kernel_ms %s parity %s'%(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d.get('parity_sample',{}).get('result')), {k:v for k,v in d.items() if 'following' in k or 'first' in k})
except Exception as e: print('dict=$d ERR', e); print(open('gpurun_out/r04/saopd_dict$d.err').read()[-1500:])
PY
done
