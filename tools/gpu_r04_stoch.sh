#!/bin/bash
cd /root/repo
python -m pytest tests -m gpu -x -q -k "stoch or round3 or subtree or prior or fuzz" 2>&1 | tail -15
for mode in closed open; do MI355PLAN_NO_TORCH=1 python tools/micro_uct_stoch.py 262144 $mode | tail -2; done
MI355PLAN_NO_TORCH=1 python tools/micro_uct_stoch.py 4096 closed | tail -1
