#!/bin/bash
# round-4: state-aware OPD, vector form of the backup's sequential half: parity, then the mix micro-benchmark; FLAGS_LIST = builds to compare
cd /root/repo
mkdir -p gpurun_out/r04
i=0
IFS=";" read -ra LIST <<< "${FLAGS_LIST:- }"
for f in "${LIST[@]}"; do
  export MP_EXTRA_FLAGS="$f"
  echo "=== build flags: [$f]"
  python -m rl_agents_amd.build > /dev/null 2>&1
  python -m pytest tests -m gpu -x -q -k "state_aware_batch" 2>&1 | tail -2
  PYTHONPATH=/root/repo python tools/micro_saopd_mix.py 16384 short 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/saopd_mix_$i.txt
  PYTHONPATH=/root/repo python tools/micro_saopd_follow.py 16384 3 2>&1 | grep -v amdgpu.ids
  i=$((i+1))
done
