#!/bin/bash
# Build a VARIANT of libmi355plan.so into build_variants/<name>/ (git-ignored, travels with the gpurun snapshot):
#   tools/build_variant.sh prof -DMP_PROFILE          -> MI355PLAN_LIB=build_variants/prof/libmi355plan.so python ...
set -e
NAME=$1; shift
OUT=/root/repo/build_variants/$NAME
mkdir -p $OUT
cd /root/repo/rl_agents_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -Wno-pass-failed -I /root/repo/include $@"
pids=()
for f in api vi uct uct_stoch opd ropd saopd; do
  # (the per-file flags of rl_agents_amd/build.py FILE_FLAGS; NO_FILE_FLAGS=1 leaves them out)
  FF=""; if [ -z "$NO_FILE_FLAGS" ] && [ $f = uct ]; then FF="-mllvm -amdgpu-sched-strategy=max-ilp"; fi
  /opt/rocm/bin/hipcc $FLAGS $FF -c $f.hip -o $OUT/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmi355plan.so $OUT/*.o -Wl,-rpath,/opt/rocm/lib
ls -la $OUT/libmi355plan.so
