#!/bin/bash
# one short, bounded GPU command (always under `timeout`)
cd /root/repo
MI355PLAN_NO_TORCH=1 timeout 200 python tools/fuzz_parity.py 12000 7071 2>&1 | tail -2
FUZZ_HEAVY=1 MI355PLAN_NO_TORCH=1 timeout 150 python tools/fuzz_parity.py 400 7072 2>&1 | tail -2
