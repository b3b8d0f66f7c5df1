#!/bin/bash
# OPD on the GPU box: parity tests, the in-kernel phase timers (build_variants/prof, -DMP_PROFILE) and the BASELINE C4 shard sizes
cd /root/repo
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_batch.py tests/test_gpu_golden.py tests/test_gpu_bench_sizes.py tests/test_gpu_variants.py tests/test_gpu_fuzz.py tests/test_gpu_env_restrictions.py -x -q -k "opd or OPD or fuzz or random or determin" 2>&1 | tail -5
if [ -f build_variants/prof/libmi355plan.so ]; then
for r in 64 1024; do
 MI355PLAN_LIB=build_variants/prof/libmi355plan.so timeout 120 python bench.py --workload opd --roots $r --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep "opd prof" | tail -1
done
fi
for r in 1024 768 256 64 8192; do
  timeout 300 python bench.py --workload opd --roots $r --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('opd roots $r', 'ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'value', d['value'], 'frac', d['roofline']['frac'])"
done
