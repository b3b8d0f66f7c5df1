#!/bin/bash
# Runs ON THE GPU BOX: unit-utilisation PMC passes (one rocprofv3 --pmc set per pass, kernel trace only) of ANY command.
#   tools/profile_units_cmd.sh <tag> <kernel substring> <command...>
# -> gpurun_out/units_<tag>.txt : per counter, the mean over the kernel's launches
set -u
TAG=$1; KSUB=$2; shift 2
OUT=/root/repo/gpurun_out/units_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
"$@" > $OUT/plain.log 2>&1
i=0
# EXTRA_SETS="A B;C D": further counter sets (one pass each), e.g. LDS / scalar-unit counters
IFS=';' read -ra EXTRA <<< "${EXTRA_SETS:-}"
for set in "${EXTRA[@]}" "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVES SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/set$i -o k -- "$@" > $OUT/set$i.log 2>&1
done
python - "$OUT" "$KSUB" <<'PY' | tee /root/repo/gpurun_out/units_$TAG.txt
import collections, csv, glob, os, sys
out, ksub = sys.argv[1], sys.argv[2]
print(open(os.path.join(out, "plain.log")).read().strip().splitlines()[-1][:300])
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "set*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print("{:34s} {:.5g}  (n={})".format(k, sum(v) / len(v), len(v)))
PY
