#!/bin/bash
# one short, bounded GPU command (always under `timeout`)
cd /root/repo
TRACE_WLS="vi_dense_exact rvi_dense_shard_exact uct_stoch" PMC_WLS="vi_dense_exact rvi_dense_shard_exact" timeout 400 bash tools/profile_gpu.sh r04 > gpurun_out/profile_r04_subset.log 2>&1
ls gpurun_out/prof_r04 | grep exact | head
