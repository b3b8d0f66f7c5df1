#!/bin/bash
# the dense-exact GPU tests + the row-block kernel with V staged piece by piece
cd /root/repo
O=gpurun_out/r04x2
mkdir -p $O
python -m pytest tests/test_gpu_vi_dense_exact.py tests/test_gpu_bench_sizes.py -m gpu -q -k "dense" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log | cut -c1-250
B="--no-cpu-baseline --headline-only --steps 5 --warmup 1"
timeout 200 python bench.py --workload vi_dense_exact $B > $O/vi_dense_exact.json 2> $O/vi_dense_exact.err
MP_VI_EXACT_V=pieces timeout 200 python bench.py --workload vi_dense_exact $B --no-parity-sample > $O/vi_dense_exact_pieces.json 2>&1
MP_VI_EXACT_WAVES=4 timeout 200 python bench.py --workload vi_dense_exact $B --no-parity-sample > $O/vi_dense_exact_w4.json 2>&1
timeout 300 python bench.py --workload rvi_dense_shard --dense-mode exact $B > $O/rvi_shard_exact.json 2> $O/rvi_shard_exact.err
MP_VI_EXACT_WAVES=4 timeout 300 python bench.py --workload rvi_dense_shard --dense-mode exact $B --no-parity-sample > $O/rvi_shard_exact_w4.json 2>&1
MP_VI_EXACT_V=global timeout 300 python bench.py --workload rvi_dense_shard --dense-mode exact $B --no-parity-sample > $O/rvi_shard_exact_global.json 2>&1
timeout 300 python bench.py --workload rvi_dense_shard --dense-mode mfma $B --no-parity-sample > $O/rvi_shard_mfma.json 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04x2/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        r = d['roofline']
        print('%-40s value %.5g kernel %s ms %.4f frac %.3f parity %s' % (f.split('/')[-1], d['value'], r.get('kernel'), r.get('kernel_ms'), r.get('frac') or -1, (d.get('parity_sample') or {}).get('result')))
    except Exception as e:
        print(f, 'ERR', e, open(f).read()[-600:])
PY
