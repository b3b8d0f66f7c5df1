#!/bin/bash
cd /root/repo
PYTHONPATH=/root/repo python tools/micro_saopd_follow.py 16384 12 2>&1 | grep -v amdgpu.ids
PYTHONPATH=/root/repo python tools/micro_saopd_follow.py 4096 16 2>&1 | grep -v amdgpu.ids
