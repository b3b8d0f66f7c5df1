#!/bin/bash
# round-4 call 4: crossover of the LDS-resident variant against the global-record variant over batch sizes
cd /root/repo
mkdir -p gpurun_out/r04
for n in 8192 16384 32768 65536 131072 524288; do
 for mode in global ldsr; do
  MP_UCT_MODEL=$mode timeout 300 python bench.py --headline-only --no-cpu-baseline --no-parity-sample --roots $n --steps 20 > gpurun_out/r04/x_${mode}_$n.json 2>/dev/null
 done
done
python - <<'PY'
import json,glob
for n in (8192,16384,32768,65536,131072,524288):
    row=[]
    for mode in ('global','ldsr'):
        try:
            d=json.loads([l for l in open('gpurun_out/r04/x_%s_%d.json'%(mode,n)) if l.startswith('{')][-1])
            row.append('%s kernel_ms %.4f value %.4g'%(mode,d['roofline']['kernel_ms'],d['value']))
        except Exception as e: row.append('%s ERR %s'%(mode,e))
    print(n,' | '.join(row))
PY
