#!/bin/bash
# one short, bounded GPU command (always under `timeout`)
cd /root/repo
timeout 200 python -X faulthandler -m pytest tests/test_gpu_batched_eval.py -m gpu -q -x -k "device_resident_loop_equals_host" 2>&1 | grep -v "Extension modules" | tail -25 | cut -c1-250
