#!/bin/bash
# full GPU suite, smoke, default bench line, 2-rank dry run (same device), batch timings
cd /root/repo
O=gpurun_out/r05c
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_full.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu_full.log
tail -30 $O/pytest_gpu_full.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
tail -3 $O/bench_default.time; tail -5 $O/bench_default.err | cut -c1-300
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/r05c/bench_default.json') if l.startswith('{')][-1])
    print('headline value %.4g ms %.3f frac %.3f traffic %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic')))
    print('general', d.get('general_model_kernel'))
    print('parity', d.get('parity_sample'))
    for k, v in d.get('workloads', {}).items():
        print('  %-24s %s' % (k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ('value', 'kernel_ms', 'frac', 'parity_sample', 'error', 'slice_seconds')}))
        for kk in ('speedup_vs_single_solve', 'vs_shared_model_kernel'):
            if kk in v: print('      ', kk, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v[kk].items() if a != 'note'})
        if isinstance(v.get('cpu_baseline'), dict): print('       cpu', {a: b for a, b in v['cpu_baseline'].items() if a in ('value', 'unit', 'cores')})
except Exception as e:
    print('bench parse failed', e)
PY
( time timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --roots 65536 > $O/bench_2rank.json 2> $O/bench_2rank.err ) 2> $O/bench_2rank.time
tail -3 $O/bench_2rank.time; tail -5 $O/bench_2rank.err | cut -c1-300
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/r05c/bench_2rank.json') if l.startswith('{')][-1])
    print('2-rank value %.4g ms %.3f' % (d['value'], d['ms_per_step']), {k: v for k, v in d['ranks'].items() if k != 'devices'})
    print('exchange', {k: v for k, v in d.get('exchange', {}).items() if k != 'note'})
    for k, v in d.get('workloads', {}).items():
        print('  %-24s %s' % (k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ('value', 'kernel_ms', 'frac', 'error', 'ranks_seen', 'cross_check', 'all_gather_ms', 'exchange', 'slice_seconds')}))
except Exception as e:
    print('2-rank parse failed', e)
PY
timeout 600 python tools/bench_batch.py > $O/bench_batch.jsonl 2> $O/bench_batch.err
grep "vi_batch" $O/bench_batch.jsonl | cut -c1-260
