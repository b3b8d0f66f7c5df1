#!/bin/bash
cd /root/repo
bash tools/profile_gpu.sh r04 > gpurun_out/profile_r04.log 2>&1
tail -5 gpurun_out/profile_r04.log
python tools/summarize_profiles.py r04 2>&1 | tail -3
ls profiles | grep r04
