#!/bin/bash
# round-4: unit counters of saopd_wave_kernel for all-light / all-heavy batches (instructions per iteration / per backup)
cd /root/repo
export PYTHONPATH=/root/repo
for w in light heavy; do
  bash tools/profile_units_cmd.sh saopd_$w saopd_wave_kernel python /root/repo/tools/micro_saopd_one.py $w 16384 2 > /dev/null 2>&1
  echo "== $w"; cat gpurun_out/units_saopd_$w.txt
done
