#!/bin/bash
# round-4 call 2: hand-written PCG64 step (v_mad_u64_u32 columns) + raw-draw thresholds: whole GPU suite (bit-exactness of
# every planner's random stream), headline and all slices
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu2.log
tail -8 gpurun_out/r04/pytest_gpu2.log
timeout 600 python bench.py > gpurun_out/r04/bench_default2.json 2> gpurun_out/r04/bench_default2.err
tail -3 gpurun_out/r04/bench_default2.err
python - <<'PY'
import json
for f in ('gpurun_out/r04/bench_default2.json',):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, 'value %.4g'%d['value'], 'ms %.3f'%d['ms_per_step'], 'frac', d['roofline'].get('frac'), 'kernel_ms', d['roofline'].get('kernel_ms'), 'parity', d.get('parity_sample',{}).get('result'), 'v4096 %.4g'%d['value_roots4096'], 'host incl %.4g'%d['value_host_inclusive'])
        for k,v in (d.get('workloads') or {}).items():
            print('  ', k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('value','kernel_ms','frac','parity_sample','error')})
    except Exception as e: print(f,'ERR',e)
PY
