// A lone wave's argmax over its 64 lanes (non-negative f64 bound + id, first maximum = lowest id): the DPP reduction of
// csrc/wave.hpp (wave_argmax_nonneg) against LDS atomics on one word pair (ds_max_u64 of the bit pattern, then ds_min_u32 of the
// ids of the lanes that hold it).  Ticks per argmax in a dependent chain (each round's values depend on the previous result).
//   hipcc --offload-arch=gfx950 -O2 -I rl_agents_amd/csrc tools/argmax_lds.hip -o build_variants/argmax_lds && build_variants/argmax_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "wave.hpp"
using namespace mp;

template <int MODE>
__global__ void k(long long *out, int n, double *res)
{
    __shared__ unsigned long long sm[2];
    const int lane = threadIdx.x;
    double acc = 0.0;
    int accid = 0;
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        // (a few lanes -inf, ties now and then)
        double u = ((lane * 37 + i * 11 + accid) & 63) < 50 ? (double)(((lane * 29 + i + accid) & 31) + 1) * 0.125 : -INFINITY;
        int id = lane + 64 * (i & 7);
        if (MODE == 0) wave_argmax_nonneg(u, id);
        if (MODE == 1) {
            typedef __attribute__((address_space(3))) unsigned long long lds64;
            typedef __attribute__((address_space(3))) unsigned lds32;
            // (inline assembly: the compiler's atomic optimizer turns a same-address atomic of a wave into a scalar loop over its lanes)
            const unsigned a0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned long long *)sm;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(u), zero = 0ull;
            unsigned long long m;
            unsigned best, big = 0xffffffffu;
            asm volatile("ds_write_b64 %0, %1\n\tds_write_b32 %0, %2 offset:8" : : "v"(a0), "v"(zero), "v"(big) : "memory");
            if (u >= 0.0) asm volatile("ds_max_u64 %0, %1" : : "v"(a0), "v"(bits) : "memory");
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(m) : "v"(a0) : "memory");
            if (u >= 0.0 && bits == m) asm volatile("ds_min_u32 %0, %1 offset:8" : : "v"(a0), "v"((unsigned)id) : "memory");
            asm volatile("ds_read_b32 %0, %1 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=v"(best) : "v"(a0) : "memory");
            id = (int)best;
            u = best == 0xffffffffu ? -INFINITY : __longlong_as_double((long long)m);
        }
        acc += u > 0 ? u : 0.0;
        accid = id;
    }
    const long long t1 = clock64();
    if (lane == 0) { out[0] = t1 - t0; res[0] = acc; res[1] = (double)accid; }
}

int main()
{
    long long *d; double *r;
    hipMalloc(&d, 16); hipMalloc(&r, 16);
    const char *names[2] = {"DPP reduction (wave_argmax_nonneg)", "LDS atomics (ds_max_u64, ds_min_u32)"};
    const int n = 4096;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, n, r);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, n, r);
            hipDeviceSynchronize();
        }
        long long h[2]; double hr[2];
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); hipMemcpy(hr, r, 16, hipMemcpyDeviceToHost);
        printf("%-40s %.1f ticks per argmax   (checksum %.3f / %d)\n", names[mode], (double)h[0] / n, hr[0], (int)hr[1]);
    }
    return 0;
}
