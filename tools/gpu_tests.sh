#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu_full.log
tail -6 gpurun_out/r04/pytest_gpu_full.log
