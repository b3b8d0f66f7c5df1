#!/bin/bash
# one short, bounded GPU command (always under `timeout`)
cd /root/repo
MI355PLAN_NO_TORCH=1 FUZZ_KINDS=vi timeout 200 python tools/fuzz_parity.py 300 5051 2>&1 | tail -2
timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -3 | cut -c1-250
