#!/bin/bash
# round-2 closing GPU pass.  Order matters: the profile passes first and their summary written into profiles/ ON THE BOX,
# so that the bench lines printed afterwards carry the HBM traffic of THIS build; everything judged is copied to gpurun_out/.
cd /root/repo
mkdir -p gpurun_out/profiles_out
bash tools/profile_gpu.sh r02 > gpurun_out/profile_r02.log 2>&1
python tools/summarize_profiles.py r02 > /dev/null 2>&1
cp profiles/r02_kernel_stats.md profiles/r02_pmc.json gpurun_out/profiles_out/
: > gpurun_out/r02_bench_lines.jsonl
for wl in uct uct_prior uct_cartpole opd ropd saopd vi rvi vi_dense rvi_dense_shard; do
  timeout 400 python bench.py --workload $wl 2> gpurun_out/r02_bench_$wl.err | grep "^{" >> gpurun_out/r02_bench_lines.jsonl
done
timeout 400 python bench.py --workload opd --roots 8192 --no-cpu-baseline 2>/dev/null | grep "^{" >> gpurun_out/r02_bench_lines.jsonl
build_variants/stream_read 4000000000 > gpurun_out/r02_stream_read_ceiling.txt 2>&1
build_variants/stream_read 25000000000 >> gpurun_out/r02_stream_read_ceiling.txt 2>&1
timeout 300 python tools/micro_vi_persist.py > gpurun_out/r02_vi_persist_ab.txt 2>&1
for v in 0 1; do echo "== MP_SAOPD_LDS=$v"; for n in 1 64 256 4096; do MP_SAOPD_LDS=$v MI355PLAN_NO_TORCH=1 timeout 200 python tools/micro_uct_opd.py saopd $n 2>&1 | grep saopd; done; done > gpurun_out/r02_saopd_lds_ab.txt 2>&1
BENCH_RCCL_STANDIN=1 timeout 300 python bench.py --workload rvi_dense_shard --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r02_shard_rccl_standin.json
bash tools/profile_units.sh uct262144 uct_kernel uct 262144 > /dev/null 2>&1
bash tools/profile_units.sh opd8192 opd_ opd 8192 > /dev/null 2>&1
bash tools/profile_units.sh opd1024 opd_ opd 1024 > /dev/null 2>&1
wc -l gpurun_out/r02_bench_lines.jsonl
