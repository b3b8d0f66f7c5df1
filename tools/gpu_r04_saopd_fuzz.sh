#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04
FUZZ_KINDS=saopd,saopd_masked MI355PLAN_NO_TORCH=1 timeout 900 python tools/fuzz_parity.py ${N:-800} 41 2>&1 | tail -5 | tee gpurun_out/r04/fuzz_saopd.txt
FUZZ_HEAVY=1 FUZZ_KINDS=saopd,saopd_masked MI355PLAN_NO_TORCH=1 timeout 900 python tools/fuzz_parity.py 150 43 2>&1 | tail -5 | tee -a gpurun_out/r04/fuzz_saopd.txt
