// Does the issue rate of a lone wavefront depend on how many of its lanes are active (and where they sit: every `stride`-th lane)?
// One wave per workgroup, `active` lanes
// run a chain of f64 fma / IEEE divisions / LDS table reads; the others leave at the top (EXEC masks them off for good).
//   hipcc --offload-arch=gfx950 -O2 tools/exec_rate.hip -o build_variants/exec_rate && build_variants/exec_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ void k(double *out, int active, int iters, int stride)
{
    __shared__ double tab[512];
    for (int i = threadIdx.x; i < 512; i += 64) tab[i] = 1.0 + i * 1e-3;
    __syncthreads();
    if ((int)threadIdx.x % stride != 0 || (int)threadIdx.x / stride >= active) return;
    double x = 1.0 + threadIdx.x * 1e-3, y = 0.5;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { x = x * 1.0000001 + 1e-9; y = y * 0.9999999 + x; }
        if (MODE == 1) { x = (x + 3.0) / (y + 2.0); y = (y + 1.5) / (x + 2.5); }
        if (MODE == 2) { const int j = (int)(x * 37.0) & 511; x = x * 0.999 + tab[j] * 1e-3; y += x; }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / iters;
    if (x + y == 12345.678) out[1] = x;
}

int main()
{
    double *d;
    hipMalloc(&d, 16);
    const char *names[3] = {"fma chain (2 fma / iter)", "IEEE division (2 / iter)", "LDS gather + fma"};
    for (int mode = 0; mode < 3; ++mode)
        for (int grid : {1, 1024})
            for (int cfg : {64, 16, 8, 4, 2, 1, 1008, 1004, 1604, 1602, 3202, 408, 416, 204, 216}) {
                const int active = cfg % 100, stride = cfg >= 100 ? cfg / 100 : 1;   // (cfg = stride * 100 + active lanes)
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, d, active, 20000, stride);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, d, active, 20000, stride);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, d, active, 20000, stride);
                    hipDeviceSynchronize();
                }
                double h[2];
                hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                printf("%-28s grid %5d  active lanes %2d stride %2d : %.1f ticks per iteration\n", names[mode], grid, active, stride, h[0]);
            }
    return 0;
}
