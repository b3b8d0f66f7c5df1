#!/bin/bash
# one short, bounded GPU command (always under `timeout`)
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_agents.py tests/test_gpu_visits.py tests/test_tree_tools.py -m gpu -q 2>&1 | tail -8 | cut -c1-250
