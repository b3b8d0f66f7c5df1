#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_round3.py tests/test_gpu_env_restrictions.py tests/test_gpu_batch.py tests/test_gpu_agents.py tests/test_gpu_variants.py -x -q > gpurun_out/r03/pytest2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03/pytest2.log
tail -15 gpurun_out/r03/pytest2.log
for cfg in "32768 4" "16384 4" "65536 4" "32768 8" "16384 8" "8192 8" "32768 2"; do
  set -- $cfg
  echo "== chunk $1 streams $2"
  MP_PIPE_CHUNK=$1 MP_PIPE_STREAMS=$2 timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'host_incl', d['value_host_inclusive'], d['host_inclusive_ms_per_step'], 'r4096', d['value_roots4096'], d['value_roots4096_host_inclusive'], d['plan_wall_ms_per_root'], d['host_inclusive_pageable_all_outputs_ms'])"
done
