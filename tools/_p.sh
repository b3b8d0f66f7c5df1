cd /root/repo
python -m pytest tests -m gpu -x -q -k "ropd or robust or fuzz or random or Robust" 2>&1 | tail -3
for loop in 0 2; do
MP_OPD_LOOP=$loop python bench.py --workload ropd --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ropd loop $loop ms', d['ms_per_step'], d['roofline'].get('kernel_ms'))"
done
