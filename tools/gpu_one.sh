#!/bin/bash
# one short, bounded GPU command (always under `timeout`)
cd /root/repo
timeout 200 python -X faulthandler -m pytest tests -m gpu -q -x -k "policy_load_errors or per_state_policies" 2>&1 | grep -v "Extension modules" | tail -5 | cut -c1-250
MI355PLAN_NO_TORCH=1 FUZZ_KINDS=uct_listed timeout 300 python tools/fuzz_parity.py 250 4043 2>&1 | tail -4
