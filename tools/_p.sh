cd /root/repo
MI355PLAN_LIB=build_variants/prof/libmi355plan.so timeout 120 python bench.py --workload opd --roots 64 --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep "prof2" | tail -5
