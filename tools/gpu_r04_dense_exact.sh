#!/bin/bash
# whole GPU suite + the dense VI kernels side by side (matrix cores / numpy's order), knobs, rocprofv3 trace and HBM counters
cd /root/repo
O=gpurun_out/r04x
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu_full.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu_full.log
tail -15 $O/pytest_gpu_full.log | cut -c1-250
B="--no-cpu-baseline --headline-only --steps 5 --warmup 1"
timeout 200 python bench.py --workload vi_dense $B > $O/vi_dense.json 2> $O/vi_dense.err
timeout 200 python bench.py --workload vi_dense_exact $B > $O/vi_dense_exact.json 2> $O/vi_dense_exact.err
MP_VI_EXACT_WAVES=8 timeout 200 python bench.py --workload vi_dense_exact $B --no-parity-sample > $O/vi_dense_exact_w8.json 2>&1
MP_VI_EXACT_NO_VLDS=1 timeout 200 python bench.py --workload vi_dense_exact $B --no-parity-sample > $O/vi_dense_exact_novlds.json 2>&1
MP_VI_EXACT_NO_VLDS=1 MP_VI_EXACT_WAVES=8 timeout 200 python bench.py --workload vi_dense_exact $B --no-parity-sample > $O/vi_dense_exact_novlds_w8.json 2>&1
timeout 200 python bench.py --workload vi_dense_exact --states 8192 $B --no-parity-sample > $O/vi_dense_exact_s8192.json 2>&1
timeout 200 python bench.py --workload vi_dense --states 8192 $B --no-parity-sample > $O/vi_dense_s8192.json 2>&1
timeout 300 python bench.py --workload rvi_dense_shard $B --no-parity-sample > $O/rvi_shard_mfma.json 2>&1
MP_VI_DENSE=exact timeout 300 python bench.py --workload rvi_dense_shard $B --no-parity-sample > $O/rvi_shard_exact.json 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04x/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        r = d['roofline']
        print('%-40s value %.5g kernel %s ms %.4f frac %.3f parity %s' % (f.split('/')[-1], d['value'], r.get('kernel'), r.get('kernel_ms'), r.get('frac') or -1, (d.get('parity_sample') or {}).get('result')))
    except Exception as e:
        print(f, 'ERR', e, open(f).read()[-600:])
PY
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace_vi_dense_exact -o vi_dense_exact -- python /root/repo/bench.py --workload vi_dense_exact $B --no-parity-sample > /root/repo/$O/trace.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /root/repo/$O/pmc_vi_dense_exact_$ctr -o vi_dense_exact -- python /root/repo/bench.py --workload vi_dense_exact --steps 3 --warmup 1 --no-cpu-baseline --headline-only --no-parity-sample > /root/repo/$O/pmc_$ctr.log 2>&1
done
cd /root/repo
head -5 $O/trace_vi_dense_exact/*/*kernel_stats.csv 2>/dev/null | cut -c1-200
python - <<'PY'
import csv, glob, collections
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg = collections.defaultdict(list)
    for f in glob.glob('gpurun_out/r04x/pmc_vi_dense_exact_%s/**/*counter_collection.csv' % ctr, recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] == ctr and 'exact' in row['Kernel_Name']:
                agg[row['Kernel_Name'][:60] + ' grid=' + row['Grid_Size']].append(float(row['Counter_Value']))
    for k, v in agg.items():
        print(ctr, k, len(v), sum(v) / len(v), 'KB per launch')
PY
