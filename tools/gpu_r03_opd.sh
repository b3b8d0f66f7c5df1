#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_batch.py tests/test_gpu_golden.py tests/test_gpu_bench_sizes.py tests/test_gpu_variants.py tests/test_gpu_fuzz.py -x -q -k "opd or OPD or fuzz or random" 2>&1 | tail -5
for r in 1024 768 256 8192; do
  timeout 300 python bench.py --workload opd --roots $r --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('opd roots $r', 'ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'value', d['value'], 'frac', d['roofline']['frac'])"
done
