#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04
MI355PLAN_NO_TORCH=1 timeout 1200 python tools/fuzz_parity.py ${N:-1200} 2024 2>&1 | tail -4 | tee gpurun_out/r04/fuzz_all.txt
FUZZ_HEAVY=1 MI355PLAN_NO_TORCH=1 timeout 900 python tools/fuzz_parity.py 300 2025 2>&1 | tail -3 | tee -a gpurun_out/r04/fuzz_all.txt
