#!/bin/bash
cd /root/repo
O=gpurun_out/r05b
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_per_episode.py tests/test_gpu_per_episode_eval.py -q > $O/pytest_per_episode.log 2>&1
echo "pytest rc=$?" >> $O/pytest_per_episode.log
tail -40 $O/pytest_per_episode.log | cut -c1-300
timeout 600 python tools/bench_batch.py > $O/bench_batch.jsonl 2> $O/bench_batch.err
tail -3 $O/bench_batch.err | cut -c1-300
grep vi_batch $O/bench_batch.jsonl | cut -c1-330
