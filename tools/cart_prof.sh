# rocprofv3 kernel time of the CartPole C3 kernel under a few settings: bash tools/cart_prof.sh ["ENV=..." ...]
export MI355PLAN_NO_TORCH=1 TMPDIR=/tmp
cd /tmp
[ $# -eq 0 ] && set -- "default" "MP_CART_FASTDIV=0" "MP_UCT_LANES=16 MP_UCT_CART_REP=0" "MP_UCT_LANES=4 MP_UCT_CART_REP=2" "MP_UCT_LANES=8 MP_UCT_CART_REP=3"
for cfg in "$@"; do
  rm -rf /tmp/cp_prof
  env $( [ "$cfg" = default ] || echo $cfg ) rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_prof -o k -- python /root/repo/tools/micro_cartpole.py 4096 > /tmp/cp_prof.log 2>&1
  echo "$cfg: $(python -c "
import csv
for r in csv.DictReader(open('/tmp/cp_prof/k_kernel_stats.csv')):
    if 'uct_kernel' in r['Name']: print('calls', r['Calls'], 'avg_us %.1f min_us %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
" 2>&1 | tail -1)"
done
