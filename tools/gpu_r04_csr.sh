#!/bin/bash
cd /root/repo
python -m pytest tests -m gpu -x -q -k "state_aware or saopd" 2>&1 | tail -3
PYTHONPATH=/root/repo python tools/micro_saopd_follow.py 16384 12 2>&1 | grep -v amdgpu.ids
PYTHONPATH=/root/repo python tools/micro_saopd_follow.py 4096 16 2>&1 | grep -v amdgpu.ids
MP_SAOPD_CSR=0 PYTHONPATH=/root/repo python tools/micro_saopd_follow.py 4096 16 2>&1 | grep -v amdgpu.ids
FUZZ_KINDS=saopd,saopd_masked MI355PLAN_NO_TORCH=1 timeout 600 python tools/fuzz_parity.py 300 51 2>&1 | tail -2
