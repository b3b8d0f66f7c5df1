#!/bin/bash
# whole GPU suite, smoke(), the default bench line (all workloads), rocprofv3 trace + HBM counters of the dense-exact workloads
cd /root/repo
O=gpurun_out/r04y
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu_full.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu_full.log
tail -6 $O/pytest_gpu_full.log | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-400
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
tail -3 $O/bench_default.time
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04y/bench_default.json') if l.startswith('{')][-1])
print('headline value %.4g ms %.3f frac %.3f traffic %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic')))
for k, v in d.get('workloads', {}).items():
    print('  %-24s %s' % (k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ('value', 'kernel_ms', 'frac', 'traffic_frac', 'parity_sample', 'error', 'slice_seconds')}))
PY
export TMPDIR=/tmp
cd /tmp
B="--no-cpu-baseline --headline-only --no-parity-sample"
for wl in vi_dense_exact rvi_dense_shard; do
  X=""; [ $wl = rvi_dense_shard ] && X="--dense-mode exact"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace_${wl}_exact -o $wl -- python /root/repo/bench.py --workload $wl $X --steps 5 --warmup 1 $B > /root/repo/$O/trace_$wl.log 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /root/repo/$O/pmc_${wl}_exact_$ctr -o $wl -- python /root/repo/bench.py --workload $wl $X --steps 3 --warmup 1 $B > /root/repo/$O/pmc_${wl}_$ctr.log 2>&1
  done
done
cd /root/repo
for f in $O/trace_*_exact/*kernel_stats.csv; do echo $f; head -4 $f | cut -c1-220; done
python - <<'PY'
import csv, glob, collections
for wl in ('vi_dense_exact', 'rvi_dense_shard'):
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        agg = collections.defaultdict(list)
        for f in glob.glob('gpurun_out/r04y/pmc_%s_exact_%s/**/*counter_collection.csv' % (wl, ctr), recursive=True):
            for row in csv.DictReader(open(f)):
                if row['Counter_Name'] == ctr and 'exact' in row['Kernel_Name']:
                    agg[row['Kernel_Name'][:60] + ' grid=' + row['Grid_Size']].append(float(row['Counter_Value']))
        for k, v in agg.items():
            print(wl, ctr, k, len(v), sum(v) / len(v), 'KB per launch')
PY
