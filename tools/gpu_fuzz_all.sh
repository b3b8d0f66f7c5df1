#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04
MI355PLAN_NO_TORCH=1 FUZZ_KINDS=vi timeout 600 python tools/fuzz_parity.py ${NVI:-500} 4041 2>&1 | tail -4 | tee gpurun_out/r04/fuzz_vi.txt
MI355PLAN_NO_TORCH=1 FUZZ_KINDS=ropd,ropd_masked timeout 600 python tools/fuzz_parity.py ${NR:-300} 4042 2>&1 | tail -4 | tee gpurun_out/r04/fuzz_ropd.txt
MI355PLAN_NO_TORCH=1 timeout 1200 python tools/fuzz_parity.py ${N:-1200} 2026 2>&1 | tail -4 | tee gpurun_out/r04/fuzz_all.txt
