// instr_rate.hip -- issue rate of the integer-multiply / f64 instructions the PCG64 step can be built from, on gfx950.
// Each kernel runs ITER x 8 independent copies of one instruction per wave (no dependency stalls), 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/instr_rate.hip -o gpurun_out/instr_rate && gpurun_out/instr_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITER 4096
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a[8], b = seed | 1u;
    uint64_t w[8];
    double d[8], e = (double)seed * 1e-9 + 1.0;
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i * 77u + seed; w[i] = a[i]; d[i] = a[i]; }
    for (int it = 0; it < ITER; ++it) {
#define MADU64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b) : "vcc");
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(e));
#define CVTU(i) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
#define ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
#define FLOOR64(i) asm volatile("v_floor_f64 %0, %0" : "+v"(d[i]));
#define MADI64(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b) : "vcc");
        if (OP == 0) { REP8(MADU64) }
        if (OP == 1) { REP8(MULLO) }
        if (OP == 2) { REP8(MULHI) }
        if (OP == 3) { REP8(MUL24) }
        if (OP == 4) { REP8(MAD24) }
        if (OP == 5) { REP8(MULHI24) }
        if (OP == 6) { REP8(FMA64) }
        if (OP == 7) { REP8(CVTU) }
        if (OP == 8) { REP8(ADDU) }
        if (OP == 9) { REP8(ADD64) }
        if (OP == 10) { REP8(FLOOR64) }
        if (OP == 11) { REP8(MADI64) }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32) ^ (uint32_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
static void run(const char *name, uint32_t *out)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 4; // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 3u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 5u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x ITER x 8 instructions
    const double cycles = ms * 1e-3 * prop.clockRate * 1e3;
    printf("%-18s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (clock %.2f GHz)\n", name, ms,
           cycles / (4.0 * ITER * 8), prop.clockRate * 1e-6);
}

int main()
{
    uint32_t *out;
    hipMalloc(&out, 256 * 4 * 1024 * sizeof(uint32_t) * 4);
    run<0>("v_mad_u64_u32", out);
    run<11>("v_mad_i64_i32", out);
    run<1>("v_mul_lo_u32", out);
    run<2>("v_mul_hi_u32", out);
    run<3>("v_mul_u32_u24", out);
    run<4>("v_mad_u32_u24", out);
    run<5>("v_mul_hi_u32_u24", out);
    run<6>("v_fma_f64", out);
    run<7>("v_cvt_u32_f64", out);
    run<10>("v_floor_f64", out);
    run<8>("v_add_u32", out);
    run<9>("v_lshl_add_u64", out);
    return 0;
}
