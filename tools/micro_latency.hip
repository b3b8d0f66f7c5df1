// micro_latency.hip -- dependent-chain latencies of the instructions the planner waves are made of (gfx950, one wavefront
// per SIMD: nothing hides an instruction's latency).  Build: hipcc --offload-arch=gfx950 -O3 tools/micro_latency.hip -o
// build_variants/micro_latency; prints cycles per chain link (s_memtime ticks / links).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../rl_agents_amd/csrc/wave.hpp"

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

#define CHAIN_KERNEL(name, decl, body)                                                     \
    __global__ void name(long long *out, double seed)                                      \
    {                                                                                      \
        decl;                                                                              \
        long long t0 = clock64();                                                          \
        for (int i = 0; i < 16; ++i) { REP64(body) }                                       \
        long long t1 = clock64();                                                          \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)sink(v0, v1, d0); }  \
    }

__device__ __forceinline__ double sink(int a, int b, double d) { return (double)a + b + d; }

CHAIN_KERNEL(k_valu_add, int v0 = threadIdx.x; int v1 = 1; double d0 = seed,
             asm volatile("v_add_u32 %0, %0, %1" : "+v"(v0) : "v"(v1));)
CHAIN_KERNEL(k_max_f64, int v0 = 0; int v1 = 1; double d0 = seed + threadIdx.x; double d1 = seed,
             asm volatile("v_max_f64 %0, %0, %1" : "+v"(d0) : "v"(d1));)
CHAIN_KERNEL(k_add_f64, int v0 = 0; int v1 = 1; double d0 = seed + threadIdx.x; double d1 = seed,
             asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(d1));)
CHAIN_KERNEL(k_mov_dpp, int v0 = threadIdx.x; int v1 = 1; double d0 = seed,
             asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v0));)
CHAIN_KERNEL(k_max_i32_dpp, int v0 = threadIdx.x; int v1 = 1; double d0 = seed,
             asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v0));)
CHAIN_KERNEL(k_max_i32_bcast, int v0 = threadIdx.x; int v1 = 1; double d0 = seed,
             asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v0));)
__global__ void k_f64_step(long long *out, double seed)
{
    double d0 = seed + threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < 16; ++i) { REP64(d0 = mp::max_step_zero<0x111>(d0);) }
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)d0; }
}
__global__ void k_argmax_nonneg(long long *out, double seed)
{
    double d0 = seed + threadIdx.x; int id = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < 1024; ++i) { double u = d0; int j = id; mp::wave_argmax_nonneg(u, j); d0 += (threadIdx.x == (j & 63)) ? -u : 0.25; }
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)d0; }
}
__global__ void k_argmax(long long *out, double seed)
{
    double d0 = seed + threadIdx.x; int id = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < 1024; ++i) { double u = d0; int j = id; mp::wave_argmax(u, j); d0 += (threadIdx.x == (j & 63)) ? -u : 0.25; }
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)d0; }
}
__global__ void k_argmax_keys(long long *out, double seed)
{
    double d0 = seed + threadIdx.x; int id = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < 1024; ++i) { double u = d0; int j = id; mp::wave_argmax_keys(u, j); d0 += (threadIdx.x == (j & 63)) ? -u : 0.25; }
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)d0; }
}
CHAIN_KERNEL(k_readlane_valu, int v0 = threadIdx.x; int v1 = 1; double d0 = seed,
             { int s; asm volatile("v_readlane_b32 %0, %1, 63" : "=s"(s) : "v"(v0)); asm volatile("v_add_u32 %0, %1, %2" : "=v"(v0) : "s"(s), "v"(v1)); })
CHAIN_KERNEL(k_cndmask, int v0 = threadIdx.x; int v1 = 1; double d0 = seed,
             asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v0) : "v"(v1) : "vcc");)
CHAIN_KERNEL(k_cmp_f64_cnd, int v0 = threadIdx.x; int v1 = 1; double d0 = seed + threadIdx.x; double d1 = seed,
             asm volatile("v_cmp_gt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(v0) : "v"(d0), "v"(d1), "v"(v1) : "vcc");)
__global__ void k_bpermute(long long *out, double seed)
{
    int v0 = threadIdx.x, v1 = ((threadIdx.x + 1) & 63) << 2;
    long long t0 = clock64();
    for (int i = 0; i < 16; ++i) { REP64(asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(v0) : "v"(v1));) }
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v0; }
}

__global__ void k_ds_read(long long *out, double seed)
{
    __shared__ int buf[256];
    buf[threadIdx.x] = (threadIdx.x * 4 + 4) & 255; buf[threadIdx.x + 64] = 0; buf[threadIdx.x + 128] = 0; buf[threadIdx.x + 192] = 0;
    __syncthreads();
    int v0 = threadIdx.x * 4;
    long long t0 = clock64();
    for (int i = 0; i < 16; ++i) { REP64(asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(v0));) }
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v0; }
}

__global__ void k_global_load(long long *out, const int *chain)
{
    int v0 = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < 1024; ++i) v0 = chain[v0];
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v0; }
}

#define RUN(name) { hipMemset(d, 0, 16); name<<<1, 64>>>(d, 1.0); hipDeviceSynchronize(); name<<<1, 64>>>(d, 1.0); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); \
    printf("%-18s %7.1f ticks per link (100 MHz memtime counter x clock ratio: see note)\n", #name, (double)h[0] / 1024.0); }

int main()
{
    long long *d, h[2];
    hipMalloc(&d, 16);
    RUN(k_valu_add) RUN(k_max_f64) RUN(k_add_f64) RUN(k_mov_dpp) RUN(k_max_i32_dpp) RUN(k_max_i32_bcast) RUN(k_f64_step) RUN(k_argmax) RUN(k_argmax_nonneg) RUN(k_argmax_keys)
    RUN(k_readlane_valu) RUN(k_cndmask) RUN(k_cmp_f64_cnd) RUN(k_bpermute) RUN(k_ds_read)
    {
        int *chain, hc[4096];
        for (int i = 0; i < 4096; ++i) hc[i] = (i * 67 + 129) & 4095;
        hipMalloc(&chain, sizeof(hc)); hipMemcpy(chain, hc, sizeof(hc), hipMemcpyHostToDevice);
        k_global_load<<<1, 64>>>(d, chain); hipDeviceSynchronize(); k_global_load<<<1, 64>>>(d, chain);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-18s %7.1f ticks per link (4-byte gather, 16 KB table: L1/L2 hits)\n", "k_global_load", (double)h[0] / 1024.0);
    }
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock rate attribute: %d kHz\n", clk);
    return 0;
}
