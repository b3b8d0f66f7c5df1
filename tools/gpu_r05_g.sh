#!/bin/bash
cd /root/repo
O=gpurun_out/r05g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_uct_quad.py tests/test_gpu_visits.py -x -q > $O/pytest_quad.log 2>&1
echo "pytest rc=$?" >> $O/pytest_quad.log
tail -12 $O/pytest_quad.log | cut -c1-250
( for q in 1 0; do MP_UCT_QUAD=$q python tools/uct_small_batch.py 1 64 4096 16384 65536; done
  for q in 1 0; do MP_UCT_QUAD=$q MI355PLAN_LIB=rl_agents_amd/lib/prof/libmi355plan.so python tools/uct_small_batch.py 1 4096; done ) 2>&1 | grep -v amdgpu.ids | tee $O/quad_timing.txt
