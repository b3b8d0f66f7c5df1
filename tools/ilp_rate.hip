// How much instruction-level parallelism does a lone wavefront turn into time?  One wave per SIMD (grid 1024) or one wave on the
// chip (grid 1); K independent dependency chains of f64 fma / IEEE divisions in one loop body, written interleaved.  If a wave
// issues dependent instructions ~8 cycles apart and independent ones ~4 apart, two chains cost what one costs.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/ilp_rate.hip -o build_variants/ilp_rate && build_variants/ilp_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, int K>
__global__ void k(double *out, int iters)
{
    double x[K], y[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { x[j] = 1.0 + threadIdx.x * 1e-3 + j; y[j] = 0.5 + j; }
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (MODE == 0) { x[j] = __fma_rn(x[j], 1.0000001, 1e-9); }
                if (MODE == 1) { x[j] = (x[j] + 3.0) / (y[j] + 2.0); }
                if (MODE == 2) { x[j] = x[j] * 1.0000001; x[j] = x[j] + 1e-9; }
            }
        }
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) s += x[j] + y[j];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / iters / 4;
    if (s == 12345.678) out[1] = s;
}

template <int MODE, int K>
static void run(double *d, const char *name)
{
    for (int grid : {1, 1024}) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<MODE, K>), dim3(grid), dim3(64), 0, 0, d, 5000); (void)hipDeviceSynchronize(); }
        double h[2];
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-26s chains %d grid %4d : %.1f ticks per link-round (%.1f per instruction-chain link)\n", name, K, grid, h[0], h[0] / K);
    }
}

int main()
{
    double *d;
    (void)hipMalloc(&d, 16);
    run<0, 1>(d, "fma"); run<0, 2>(d, "fma"); run<0, 3>(d, "fma"); run<0, 4>(d, "fma");
    run<2, 1>(d, "mul + add"); run<2, 2>(d, "mul + add"); run<2, 4>(d, "mul + add");
    run<1, 1>(d, "add + IEEE division"); run<1, 2>(d, "add + IEEE division"); run<1, 3>(d, "add + IEEE division"); run<1, 4>(d, "add + IEEE division");
    return 0;
}
