#!/bin/bash
# Runs ON THE GPU BOX: PMC passes of uct_kernel<5, ENV_TABLE> for the two tree layouts (MP_UCT_TREE) at 262 144 and 4 096
# roots -> gpurun_out/uct_layout/<layout>_<roots>_<set>/ ; tools/summarize_uct_layout.py writes profiles/r02_uct_tree_layout.md
set -u
OUT=/root/repo/gpurun_out/uct_layout
mkdir -p $OUT
export TMPDIR=/tmp MI355PLAN_NO_TORCH=1
cd /tmp
for lay in rootmajor interleaved group; do
  for n in 262144 4096; do
    MP_UCT_TREE=$lay python /root/repo/tools/micro_uct_opd.py uct $n > $OUT/${lay}_${n}_plain.log 2>&1
    i=0
    for set in "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
      i=$((i+1))
      MP_UCT_TREE=$lay timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${lay}_${n}_set$i -o uct -- \
          python /root/repo/tools/micro_uct_opd.py uct $n > $OUT/${lay}_${n}_set$i.log 2>&1
    done
  done
done
find $OUT -name "*counter_collection.csv" | wc -l
