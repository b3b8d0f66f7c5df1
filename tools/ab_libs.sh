# A/B of library builds on ONE box: bash tools/ab_libs.sh "<workload args>;<workload args>;..." lib1 lib2 ...   ("tree" = the in-tree library)
IFS=';' read -ra WLS <<< "$1"; shift
for rep in 1 2; do
for wl in "${WLS[@]}"; do
  for lib in "$@"; do
    if [ $lib = tree ]; then unset MI355PLAN_LIB; else export MI355PLAN_LIB=/root/repo/build_variants/$lib/libmi355plan.so; fi
    python bench.py --workload $wl --headline-only --no-cpu-baseline --no-parity-sample 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s %-8s ms/step %.4f kernel_ms %.4f' % ('$wl', '$lib', d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done
done
