#!/bin/bash
# per-launch counters of saopd_wave_kernel over a first plan and two following plans
cd /root/repo
export PYTHONPATH=/root/repo TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r04/follow_units
mkdir -p $OUT
cd /tmp
python /root/repo/tools/micro_saopd_follow.py 16384 3
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TA_BUSY_avr GRBM_GUI_ACTIVE SQ_INSTS_LDS" "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/set$i -o k -- python /root/repo/tools/micro_saopd_follow.py 16384 3 > $OUT/set$i.log 2>&1
done
python - "$OUT" <<'PY' | tee /root/repo/gpurun_out/r04/saopd_follow_units.txt
import collections, csv, glob, os, sys
out = sys.argv[1]
rows = collections.defaultdict(dict)
for f in glob.glob(os.path.join(out, "set*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "saopd_wave_kernel" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].setdefault(int(r["Dispatch_Id"]), 0.0)
            rows[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
for k in sorted(rows):
    v = [rows[k][d] for d in sorted(rows[k])]
    print("{:22s} per launch: {}".format(k, "  ".join("%.4g" % x for x in v)))
PY
