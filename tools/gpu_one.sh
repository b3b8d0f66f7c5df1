#!/bin/bash
cd /root/repo
for L in 64 32 16 8 4 2 1; do
  echo "== cartpole lanes $L"
  MP_UCT_LANES=$L python bench.py --workload uct_cartpole --steps 10 --warmup 2 --no-cpu-baseline --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('  value %.4g kernel_ms %.4f parity %s' % (d['value'], r['kernel_ms'], (d.get('parity_sample') or {}).get('result')))"
done
for L in 64 16 8 4; do
  echo "== uct 4096 roots lanes $L"
  MP_UCT_LANES=$L python bench.py --workload uct --roots 4096 --steps 10 --warmup 2 --no-cpu-baseline --headline-only --no-parity-sample 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('  value %.4g kernel_ms %.4f' % (d['value'], r['kernel_ms']))"
done
