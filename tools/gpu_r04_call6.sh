#!/bin/bash
# round-4 call 6: f-3 (tree consumers, ownership), policy type random on listing orders: whole GPU suite
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu6.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu6.log
tail -40 gpurun_out/r04/pytest_gpu6.log
