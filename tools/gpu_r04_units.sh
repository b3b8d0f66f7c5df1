#!/bin/bash
# unit counters of the LDS-resident UCT kernel and of the record-gather kernel at the headline batch (one launch each: no chunking)
cd /root/repo
export MP_PIPE_CHUNK=0
export EXTRA_SETS="SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA;SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE;SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH;SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"
MP_UCT_MODEL=ldsr bash tools/profile_units_cmd.sh r04_uct_ldsr uct_kernel python /root/repo/tools/micro_uct_opd.py uct 262144 > /dev/null 2>&1
MP_UCT_MODEL=global bash tools/profile_units_cmd.sh r04_uct_global uct_kernel python /root/repo/tools/micro_uct_opd.py uct 262144 > /dev/null 2>&1
echo ---- ldsr; cat gpurun_out/units_r04_uct_ldsr.txt
echo ---- global; cat gpurun_out/units_r04_uct_global.txt
grep -l "rror" gpurun_out/units_r04_uct_ldsr/set*.log | head
