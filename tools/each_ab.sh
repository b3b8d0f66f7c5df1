# per-root-model batches (uct_per_root_model slice): the default routing against four roots per wavefront (MP_UCT_ROW=1) and a wavefront per root (MP_UCT_ROW=0)
cd /root/repo
for n in 256 512 1024 2048 4096; do
 for e in default 1 0; do
  if [ $e = default ]; then unset MP_UCT_EACH MP_UCT_ROW; else unset MP_UCT_EACH; export MP_UCT_ROW=$e; fi
  python bench.py --workload uct_per_root_model --roots $n --headline-only --no-cpu-baseline --no-parity-sample 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('roots %5d  MP_UCT_ROW=%-7s %-16s ms/step %.4f kernel_ms %.4f' % ($n, '$e', d['roofline'].get('kernel_variant'), d['ms_per_step'], d['roofline']['kernel_ms']))"
 done
done
