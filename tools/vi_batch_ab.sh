# cluster form vs one workgroup per MDP, S = 10 000: bash tools/vi_batch_ab.sh "N K" ...   (K = 0: cluster form off)
for cfg in "$@"; do set -- $cfg
  MP_VI_BATCH_CLUSTER=$2 python bench.py --workload vi_batch --states 10000 --roots $1 --headline-only --no-cpu-baseline 2>/dev/null | grep '^{' | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); d=d.get('record', d); r=d['roofline']
print('N $1 K $2', r.get('kernel'), 'ms/step %.4f kernel_ms %.4f value %.4g frac %.3f par %s' % (d['ms_per_step'], r['kernel_ms'], d['value'], r['frac'], d.get('parity_sample')), 'ratio %.1f' % d['speedup_vs_single_solve']['ratio'] if 'speedup_vs_single_solve' in d else '')"
done
