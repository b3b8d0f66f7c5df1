#!/bin/bash
# whole GPU suite + the state-aware bench line
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu_full.log
tail -6 gpurun_out/r04/pytest_gpu_full.log
timeout 300 python bench.py --workload saopd --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r04/saopd_bench.json 2> gpurun_out/r04/saopd_bench.err
python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r04/saopd_bench.json') if l.startswith('{')][-1])
    print('bench saopd: value %.4g ms %.3f kernel_ms %s parity %s first+following %s'%(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d.get('parity_sample',{}).get('result'), d['config'].get('kernel_ms_first_and_following_plans')))
except Exception as e: print('ERR', e); print(open('gpurun_out/r04/saopd_bench.err').read()[-1500:])
PY
