#!/bin/bash
# round-3 call 1: the whole GPU suite (new: distributed, dense |A|=5, persist fallback), 2-rank dry run of bench.py on the
# one GPU, OPD / ROPD lines with measured algorithmic bytes
cd /root/repo
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r03/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03/pytest_gpu.log
tail -30 gpurun_out/r03/pytest_gpu.log
for wl in opd ropd; do
  timeout 300 python bench.py --workload $wl 2> gpurun_out/r03/bench_$wl.err | grep "^{" > gpurun_out/r03/bench_$wl.json
done
timeout 300 python bench.py --workload opd --roots 8192 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r03/bench_opd8192.json
BENCH_SAME_DEVICE=1 BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 1 --roots 65536 --no-cpu-baseline > gpurun_out/r03/bench_2rank_dry.log 2>&1
timeout 300 python bench.py > gpurun_out/r03/bench_uct.json 2> gpurun_out/r03/bench_uct.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('traffic_frac'), d['roofline'].get('kernel_ms'))
    except Exception as e: print(f, 'ERR', e)
PY
grep -o '"ranks": {.*' gpurun_out/r03/bench_2rank_dry.log | cut -c1-400
