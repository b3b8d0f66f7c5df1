#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04z
( time python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r04z/bench_2rank.json 2> gpurun_out/r04z/bench_2rank.err ) 2>&1 | tail -3
echo "rc=$?"
tail -c 1500 gpurun_out/r04z/bench_2rank.json
tail -5 gpurun_out/r04z/bench_2rank.err
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r04z/bench_2rank_tr.json 2> gpurun_out/r04z/bench_2rank_tr.err ) 2>&1 | tail -3
python - <<'PY'
import json
for f in ('gpurun_out/r04z/bench_2rank.json', 'gpurun_out/r04z/bench_2rank_tr.json'):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['n_gpus'], d['value'], d['ranks'], d['scaling'])
    except Exception as e:
        print(f, 'ERR', e)
PY
