#!/bin/bash
# usage: tools/gpu.sh <timeout-seconds> <script under tools/> [log]   -- builds the library HERE first (a stale .so travels), then gpurun
set -e
cd /root/repo
python -m rl_agents_amd.build > /tmp/build.log 2>&1 || { tail -30 /tmp/build.log; exit 1; }
python -c "from rl_agents_amd import native; native.load()" || exit 1
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "bash tools/$2"
