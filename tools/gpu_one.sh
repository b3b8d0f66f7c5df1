#!/bin/bash
cd /root/repo
MI355PLAN_LIB=build_variants/prof/libmi355plan.so timeout 300 python bench.py --workload saopd --no-cpu-baseline --no-parity-sample --steps 1 --warmup 0 2>/dev/null | grep "saopd prof" | tail -8
