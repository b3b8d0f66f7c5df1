#!/bin/bash
cd /root/repo
O=gpurun_out/r05d
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_full.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu_full.log
tail -15 $O/pytest_gpu_full.log | cut -c1-300
timeout 300 python bench.py --workload vi_batch --states 10000 --roots 64 --steps 5 --warmup 2 > $O/vi_batch_64.json 2> $O/vi_batch_64.err
timeout 300 python bench.py --workload vi_batch --states 10000 --roots 256 --steps 5 --warmup 2 > $O/vi_batch_256.json 2>> $O/vi_batch_64.err
python - <<'PY'
import json
for f in ('vi_batch_64', 'vi_batch_256'):
    try:
        d = json.loads([l for l in open('gpurun_out/r05d/%s.json' % f) if l.startswith('{')][-1])
        print(f, 'kernel_ms %.3f value %.4g frac %.3f' % (d['roofline']['kernel_ms'], d['value'], d['roofline']['frac']), d['speedup_vs_single_solve']['ratio'], d['parity_sample']['result'])
    except Exception as e:
        print(f, 'failed', e)
PY
