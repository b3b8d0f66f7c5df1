#!/bin/bash
# round-2 GPU call 1: bench-size parity tests, counter calibration, new bench fields
cd /root/repo
mkdir -p gpurun_out/calib_r02
(free -g; echo; cat /sys/fs/cgroup/memory.max; nproc; cat /sys/fs/cgroup/cpu.max) > gpurun_out/r2_host.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_bench_sizes.py -x -q --durations=10 > gpurun_out/r2_sizes.log 2>&1
echo "sizes rc=$?" >> gpurun_out/r2_sizes.log
timeout 120 build_variants/gather_calib 25 > gpurun_out/calib_r02/plain.log 2>&1
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/calib_r02/fetch -o calib -- /root/repo/build_variants/gather_calib 25 > /root/repo/gpurun_out/calib_r02/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /root/repo/gpurun_out/calib_r02/write -o calib -- /root/repo/build_variants/gather_calib 25 > /root/repo/gpurun_out/calib_r02/write.log 2>&1
cd /root/repo
timeout 300 python bench.py > gpurun_out/r2_bench_uct.json 2> gpurun_out/r2_bench_uct.err
timeout 300 python bench.py --workload uct_prior > gpurun_out/r2_bench_uct_prior.json 2> gpurun_out/r2_bench_uct_prior.err
tail -5 gpurun_out/r2_sizes.log; cat gpurun_out/calib_r02/plain.log; cat gpurun_out/r2_bench_uct.json | head -c 3000; tail -3 gpurun_out/r2_bench_uct.err
