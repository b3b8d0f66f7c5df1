#!/bin/bash
# round-3 profile pass: rocprofv3 kernel stats + PMC passes (tools/profile_gpu.sh), the summaries written into profiles/
# ON THE BOX so that the bench lines printed afterwards carry the HBM traffic of THIS build; everything judged is copied to
# gpurun_out/.
cd /root/repo
mkdir -p gpurun_out/profiles_out gpurun_out/r03
bash tools/profile_gpu.sh r03 > gpurun_out/profile_r03.log 2>&1
python tools/summarize_profiles.py r03 > /dev/null 2>&1
cp profiles/r03_kernel_stats.md profiles/r03_pmc.json gpurun_out/profiles_out/
: > gpurun_out/r03_bench_lines.jsonl
for wl in uct uct_prior uct_cartpole uct_stoch opd ropd saopd vi rvi vi_dense rvi_dense_shard; do
  timeout 400 python bench.py --workload $wl 2> gpurun_out/r03/bench_$wl.err | grep "^{" >> gpurun_out/r03_bench_lines.jsonl
done
timeout 400 python bench.py --workload opd --roots 8192 --no-cpu-baseline 2>/dev/null | grep "^{" >> gpurun_out/r03_bench_lines.jsonl
timeout 300 python tools/eval_fps.py 4096 > gpurun_out/r03_batched_eval_fps.txt 2>&1
MI355PLAN_NO_TORCH=1 timeout 300 python tools/micro_host_path.py > gpurun_out/r03_host_path.txt 2>&1
BENCH_RCCL_STANDIN=1 timeout 300 python bench.py --workload rvi_dense_shard --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r03_shard_rccl_standin.json
BENCH_SAME_DEVICE=1 BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 1 --roots 65536 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r03_bench_2rank_same_device.json
wc -l gpurun_out/r03_bench_lines.jsonl; cat gpurun_out/r03_batched_eval_fps.txt
python - <<'PY'
import json
for l in open('gpurun_out/r03_bench_lines.jsonl'):
    d=json.loads(l); r=d['roofline']
    print(d['config']['workload'][:60], 'value %.4g'%d['value'], 'ms %.4g'%d['ms_per_step'], 'kernel_ms %.4g'%r['kernel_ms'], 'frac %.3f'%r['frac'], 'traffic_frac', r.get('traffic_frac'))
PY
