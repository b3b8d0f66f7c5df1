#!/bin/bash
# one short, bounded GPU command (always under `timeout`)
cd /root/repo
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -4 | cut -c1-250
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
timeout 200 python bench.py --workload uct_stoch --steps 5 --warmup 1 --no-cpu-baseline --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('uct_stoch closed kernel_ms %.4f parity %s' % (r['kernel_ms'], (d.get('parity_sample') or {}).get('result')))"
