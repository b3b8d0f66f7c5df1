# A/B of uct_lone_kernel builds on one box: the in-tree library against build_variants/<name> (bash tools/lone_ab.sh <name>)
cd /root/repo
V=${1:-base}
timeout 900 python -m pytest tests/test_gpu_uct_lone.py tests/test_gpu_uct.py tests/test_gpu_per_episode.py -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
for rep in 1 2; do
for lib in tree $V; do
  if [ $lib = tree ]; then unset MI355PLAN_LIB; else export MI355PLAN_LIB=/root/repo/build_variants/$lib/libmi355plan.so; fi
  MI355PLAN_AB=1 python tools/uct_small_batch.py 1 2 8 64 256 2>/dev/null | sed "s/^/$lib /"
done
done
unset MI355PLAN_LIB
python tools/agent_latency.py 2>&1 | grep MCTS
