#!/bin/bash
cd /root/repo
python -m pytest tests -m gpu -x -q -k "stoch" 2>&1 | tail -3
for mode in closed open; do MI355PLAN_NO_TORCH=1 PYTHONPATH=/root/repo python tools/micro_uct_stoch.py 262144 $mode | tail -2; done
MI355PLAN_NO_TORCH=1 PYTHONPATH=/root/repo python tools/micro_uct_stoch.py 65536 closed | tail -1
