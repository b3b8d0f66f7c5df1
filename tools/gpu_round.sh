#!/bin/bash
# What one round's GPU check runs ON THE GPU BOX (through gpurun --timeout 3000 -- 'bash tools/gpu_round.sh [tag]'):
# the whole GPU suite, smoke(), the default bench line (headline + every slice), the 2-rank dry run of `--gpus 2` on the one
# device, and -- with PROFILE=1 -- the rocprofv3 passes of tools/profile_gpu.sh.  Everything lands under gpurun_out/<tag>/.
TAG=${1:-round}
cd /root/repo
O=gpurun_out/$TAG
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_full.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu_full.log
tail -8 $O/pytest_gpu_full.log | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-160
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
tail -3 $O/bench_default.time
( time timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --roots 65536 > $O/bench_2rank.json 2> $O/bench_2rank.err ) 2> $O/bench_2rank.time
python - $O <<'PY'
import json, sys
o = sys.argv[1]
for name in ("bench_default", "bench_2rank"):
    try:
        d = json.loads([l for l in open("%s/%s.json" % (o, name)) if l.startswith("{")][-1])
    except Exception as e:
        print(name, "no JSON line:", e)
        continue
    print(name, "value %.4g %s, %.3f ms/step, frac %.3f, parity %s" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d.get("parity_sample")), "line bytes", len(json.dumps(d, separators=(",", ":"))))
    if "exchange" in d:
        print("   exchange", {k: v for k, v in d["exchange"].items() if k != "note"}, {k: v for k, v in d["ranks"].items() if k in ("ranks_seen", "cross_check", "backend")})
    for k, v in d.get("workloads", {}).items():
        print("   %-24s %s" % (k, v))
PY
[ -n "$PROFILE" ] && bash tools/profile_gpu.sh $TAG > $O/profile.log 2>&1
true
