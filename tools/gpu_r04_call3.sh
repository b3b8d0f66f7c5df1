#!/bin/bash
# round-4 call 3: LDS-resident model variant of the UCT kernel (MP_UCT_MODEL=ldsr): parity suite under the variant, A/B timing
cd /root/repo
mkdir -p gpurun_out/r04
MP_UCT_MODEL=ldsr python -m pytest tests -m gpu -x -q -k "uct or mcts or golden or fuzz or agents or variants or bench_sizes or cartpole" > gpurun_out/r04/pytest_ldsr.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04/pytest_ldsr.log
tail -12 gpurun_out/r04/pytest_ldsr.log
for mode in global ldsr; do
  MP_UCT_MODEL=$mode timeout 300 python bench.py --headline-only --no-cpu-baseline > gpurun_out/r04/bench_uct_$mode.json 2> gpurun_out/r04/bench_uct_$mode.err
  tail -2 gpurun_out/r04/bench_uct_$mode.err
done
for w in 1 2 4; do
  MP_UCT_MODEL=ldsr MP_UCT_LDSR_WAVES=$w timeout 300 python bench.py --headline-only --no-cpu-baseline --roots 4096 --steps 50 > gpurun_out/r04/bench_uct_ldsr_4096_w$w.json 2>/dev/null
done
MP_UCT_MODEL=global timeout 300 python bench.py --headline-only --no-cpu-baseline --roots 4096 --steps 50 > gpurun_out/r04/bench_uct_global_4096.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/bench_uct_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, 'value %.4g'%d['value'], 'ms %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%d['roofline'].get('kernel_ms'), 'parity', d.get('parity_sample',{}).get('result'), 'v4096 %.4g'%d['value_roots4096'], 'lat', {k:round(v,4) for k,v in d['config']['latency'].items() if 'kernel_ms' in k})
    except Exception as e: print(f,'ERR',e)
PY
