#!/bin/bash
cd /root/repo
O=gpurun_out/r04x4
mkdir -p $O
python -m pytest tests/test_gpu_vi_dense_exact.py -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log | cut -c1-250
B="--no-cpu-baseline --headline-only --steps 5 --warmup 1 --no-parity-sample"
for S in 10000 20000 30000; do
  timeout 200 python bench.py --workload vi_dense_exact --states $S $B > $O/exact_S$S.json 2>&1
done
MP_VI_EXACT_V=pieces timeout 200 python bench.py --workload vi_dense_exact --states 10000 $B > $O/exact_S10000_pieces.json 2>&1
timeout 300 python bench.py --workload rvi_dense_shard --dense-mode exact $B > $O/shard_exact.json 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04x4/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        r = d['roofline']
        print('%-44s kernel %-40s ms %.4f frac %.3f' % (f.split('/')[-1], r.get('kernel'), r.get('kernel_ms'), r.get('frac') or -1))
    except Exception as e:
        print(f, 'ERR', e, open(f).read()[-400:])
PY
