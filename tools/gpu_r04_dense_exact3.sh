#!/bin/bash
# where does the numpy-order dense kernel lose against the matrix-core one on long rows?  (row length, models, waves)
cd /root/repo
O=gpurun_out/r04x3
mkdir -p $O
B="--no-cpu-baseline --headline-only --steps 5 --warmup 1 --no-parity-sample"
for S in 14000 20000 30000; do
  timeout 200 python bench.py --workload vi_dense_exact --states $S $B > $O/exact_S$S.json 2>&1
  timeout 200 python bench.py --workload vi_dense --states $S $B > $O/mfma_S$S.json 2>&1
done
MP_VI_EXACT_V=global timeout 200 python bench.py --workload vi_dense_exact --states 20000 $B > $O/exact_S20000_global.json 2>&1
MP_VI_EXACT_WAVES=16 MP_VI_EXACT_V=global timeout 200 python bench.py --workload vi_dense_exact --states 20000 $B > $O/exact_S20000_global_w16.json 2>&1
timeout 300 python bench.py --workload rvi_dense_shard --states 10000 --roots 10000 --dense-mode exact $B > $O/shard_S10000_rows10000_exact.json 2>&1
timeout 300 python bench.py --workload rvi_dense_shard --states 10000 --roots 10000 --dense-mode mfma $B > $O/shard_S10000_rows10000_mfma.json 2>&1
timeout 300 python bench.py --workload rvi_dense_shard --states 50000 --roots 1250 --dense-mode exact $B > $O/shard_S50000_rows1250_exact.json 2>&1
timeout 300 python bench.py --workload rvi_dense_shard --states 50000 --roots 1250 --dense-mode mfma $B > $O/shard_S50000_rows1250_mfma.json 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04x3/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        r = d['roofline']
        print('%-44s kernel %-40s ms %.4f frac %.3f' % (f.split('/')[-1], r.get('kernel'), r.get('kernel_ms'), r.get('frac') or -1))
    except Exception as e:
        print(f, 'ERR', e, open(f).read()[-400:])
PY
