#!/bin/bash
# round 5, first GPU call: the per-episode-model tests, then the timings of the batch paths
cd /root/repo
O=gpurun_out/r05a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_per_episode.py -x -q > $O/pytest_per_episode.log 2>&1
echo "pytest rc=$?" >> $O/pytest_per_episode.log
tail -25 $O/pytest_per_episode.log | cut -c1-300
timeout 600 python tools/bench_batch.py > $O/bench_batch.jsonl 2> $O/bench_batch.err
tail -3 $O/bench_batch.err | cut -c1-300
cat $O/bench_batch.jsonl | cut -c1-400
