#!/bin/bash
cd /root/repo
python -m pytest tests -m gpu -x -q -k "action_counts or restricted_actions_batch or uct" 2>&1 | tail -4
FUZZ_KINDS=uct_policy,uct_listed,uct MI355PLAN_NO_TORCH=1 timeout 600 python tools/fuzz_parity.py 400 77 2>&1 | tail -3
