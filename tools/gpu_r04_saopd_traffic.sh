#!/bin/bash
# HBM traffic of the FIRST plan of fresh planners only (bench saopd's timed launch), two --pmc passes
cd /root/repo
export PYTHONPATH=/root/repo TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r04/saopd_traffic
mkdir -p $OUT
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$ctr -o k -- python /root/repo/tools/micro_saopd_one.py mix 16384 4 > $OUT/$ctr.log 2>&1
done
python - "$OUT" <<'PY' | tee /root/repo/gpurun_out/r04/saopd_first_plan_traffic.txt
import collections, csv, glob, os, sys
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "saopd_wave_kernel" in r["Kernel_Name"]:
            agg[(r["Counter_Name"], int(r["Dispatch_Id"]))].append(float(r["Counter_Value"]))
per = collections.defaultdict(list)
for (c, d), v in agg.items():
    per[c].append(sum(v))
for c in sorted(per):
    print(c, "per launch (raw counter units, summed over XCDs):", ["%.4g" % x for x in per[c]])
PY
tail -2 $OUT/FETCH_SIZE.log
