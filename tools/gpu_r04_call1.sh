#!/bin/bash
# round-4 call 1: GPU suite (new: device-resident sharded exchange), default bench line with all workload slices,
# plain `python bench.py --gpus 2` (self-launch; dry run on the one GPU)
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r04/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu.log
tail -25 gpurun_out/r04/pytest_gpu.log
( time timeout 600 python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err ) 2>&1 | tail -3
tail -5 gpurun_out/r04/bench_default.err
timeout 400 python bench.py --gpus 2 --steps 5 --warmup 1 --roots 65536 --no-cpu-baseline > gpurun_out/r04/bench_2rank_dry.json 2> gpurun_out/r04/bench_2rank_dry.err
echo "2rank rc=$?"
tail -5 gpurun_out/r04/bench_2rank_dry.err
python - <<'PY'
import json
for f in ('gpurun_out/r04/bench_default.json','gpurun_out/r04/bench_2rank_dry.json'):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, 'value %.4g'%d['value'], 'ms %.3f'%d['ms_per_step'], 'frac', d['roofline'].get('frac'), 'kernel_ms', d['roofline'].get('kernel_ms'), 'parity', d.get('parity_sample',{}).get('result'))
        print('  ranks', {k:v for k,v in d['ranks'].items() if k!='devices'})
        for k,v in (d.get('workloads') or {}).items():
            print('  ', k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk not in ('workload','parity_detail','kernel')})
    except Exception as e: print(f,'ERR',e)
PY
