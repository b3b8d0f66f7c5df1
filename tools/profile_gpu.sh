#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace statistics and HBM-traffic PMC passes of the
# bench workloads, written under gpurun_out/prof_$TAG/ (scratch); tools/summarize_profiles.py turns them
# into the tracked profiles/$TAG_*.md summaries.
#   usage: tools/profile_gpu.sh r01
set -u
TAG=${1:-r05}
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# "opd8192" = --workload opd --roots 8192 (BASELINE C4 on one GPU: the high-occupancy kernel)
args_of() {
  if [ "$1" = opd8192 ]; then echo "--workload opd --roots 8192";
  elif [ "$1" = rvi_dense_shard ]; then echo "--workload rvi_dense_shard --dense-mode mfma";
  elif [ "$1" = rvi_dense_shard_exact ]; then echo "--workload rvi_dense_shard --dense-mode exact";
  elif [ "$1" = uct4096 ]; then echo "--workload uct --roots 4096";
  elif [ "$1" = uct1 ]; then echo "--workload uct --roots 1";
  elif [ "$1" = uct256 ]; then echo "--workload uct --roots 256";
  elif [ "$1" = uct1024 ]; then echo "--workload uct --roots 1024";
  elif [ "$1" = vi_batch ]; then echo "--workload vi_batch --states 120 --roots 4096";
  elif [ "$1" = vi_batch_s10000 ]; then echo "--workload vi_batch --states 10000 --roots 64";
  elif [ "$1" = vi_batch_s10000_256 ]; then echo "--workload vi_batch --states 10000 --roots 256";
  else echo "--workload $1"; fi; }
# (TRACE_WLS / PMC_WLS: re-collect a subset after a kernel changed; summarize_profiles.py reads whatever is there)
for wl in ${TRACE_WLS:-uct uct4096 uct1024 uct256 uct1 uct_per_root_model uct_prior uct_cartpole uct_stoch opd opd8192 ropd saopd vi rvi vi_batch vi_batch_s10000 vi_batch_s10000_256 vi_dense vi_dense_exact rvi_dense_shard rvi_dense_shard_exact}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl -o $wl -- \
      python /root/repo/bench.py $(args_of $wl) --steps 5 --warmup 1 --no-cpu-baseline --headline-only --no-parity-sample > $OUT/trace_$wl.log 2>&1
done
# HBM traffic: FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots) -> two runs each; counters only,
# no tracing domains besides the kernel trace
for wl in ${PMC_WLS:-uct uct4096 uct_per_root_model uct_prior uct_cartpole uct_stoch vi rvi vi_batch vi_batch_s10000 vi_batch_s10000_256 vi_dense vi_dense_exact rvi_dense_shard rvi_dense_shard_exact opd opd8192 ropd saopd}; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_${wl}_$ctr -o $wl -- \
        python /root/repo/bench.py $(args_of $wl) --steps 3 --warmup 1 --no-cpu-baseline --headline-only --no-parity-sample > $OUT/pmc_${wl}_$ctr.log 2>&1
  done
done
find $OUT -name "*.csv" | head -50
