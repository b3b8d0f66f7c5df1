#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04
PYTHONPATH=/root/repo python tools/micro_saopd_mix.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/saopd_mix.txt
