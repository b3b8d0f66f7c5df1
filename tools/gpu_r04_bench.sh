#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r04
( time timeout 900 python bench.py > gpurun_out/r04/bench_default7.json 2> gpurun_out/r04/bench_default7.err ) 2>&1 | tail -3
tail -3 gpurun_out/r04/bench_default7.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_default7.json') if l.startswith('{')][-1])
r=d['roofline']
print('value %.4g'%d['value'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'traffic', r.get('traffic'), 'committed', r.get('traffic_committed'), 'tfrac', r.get('traffic_frac'))
print(r.get('traffic_counters'), r.get('traffic_live'))
print({k:v.get('parity_sample') for k,v in d['workloads'].items()})
PY
