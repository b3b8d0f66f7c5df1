#!/bin/bash
# one short, bounded GPU command (always under `timeout`)
cd /root/repo
export MP_PIPE_CHUNK=0
export EXTRA_SETS="SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"
POLICY=1 timeout 500 bash tools/profile_units_cmd.sh r04_uct_prior uct_kernel python /root/repo/tools/micro_uct_opd.py uct 262144 > /dev/null 2>&1
cat gpurun_out/units_r04_uct_prior.txt
