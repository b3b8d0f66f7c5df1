#!/bin/bash
# one short, bounded GPU command (always under `timeout`)
cd /root/repo
timeout 300 python -X faulthandler -m pytest tests/test_gpu_visits.py tests/test_gpu_batch.py -m gpu -q -k "visits or policy" 2>&1 | grep -v "Extension modules" | tail -40 | cut -c1-300
