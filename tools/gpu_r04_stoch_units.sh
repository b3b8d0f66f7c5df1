#!/bin/bash
cd /root/repo
export EXTRA_SETS="SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA;SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU"
for mode in closed open; do
  bash tools/profile_units_cmd.sh r04_stoch_$mode uct_stoch_kernel python /root/repo/tools/micro_uct_stoch.py 262144 $mode > /dev/null 2>&1
  echo ---- $mode; cat gpurun_out/units_r04_stoch_$mode.txt
done
