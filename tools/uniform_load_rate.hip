// uniform_load_rate.hip -- what a wave-uniform read costs on gfx950, in the regime the state-aware planner runs in
// (8 waves per SIMD, every wave a chain of DEPENDENT reads of one small L2-resident array: a pointer chase).
//   mode 0: every lane loads the same address, as C++ (becomes an s_load); mode 4: the same as a forced global_load_dword
//   mode 1: lane 0 loads, v_readfirstlane broadcasts (EXEC = 1 lane)
//   mode 2: scalar load (s_load_dword through the scalar cache; only valid for data nobody writes)
//   mode 3: scalar load with glc (bypasses the scalar cache: coherent with vector stores of this kernel)
// prints ns per dependent read per wave and reads per second for the whole chip.
//   hipcc --offload-arch=gfx950 -O3 tools/uniform_load_rate.hip -o build_variants/uniform_load_rate
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define STEPS 20000

template <int MODE>
__global__ __launch_bounds__(64, 8) void chase(const int32_t *next, int32_t *out, int n)
{
    int i = (blockIdx.x * 97) % n;
    const int lane = threadIdx.x;
    for (int s = 0; s < STEPS; ++s) {
        if (MODE == 0) {
            i = next[i]; // (the compiler proves the address uniform and the array read-only: it emits an s_load itself)
        } else if (MODE == 4) { // a real vector load by all 64 lanes of the same address
            const int32_t *p = next + i;
            int v;
            asm volatile("global_load_dword %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(0), "s"(p) : "memory");
            i = __builtin_amdgcn_readfirstlane(v);
        } else if (MODE == 1) {
            int v = 0;
            if (lane == 0) v = next[i];
            i = __builtin_amdgcn_readfirstlane(v);
        } else if (MODE == 2) {
            typedef const int32_t __attribute__((address_space(4))) *cptr;
            i = ((cptr)(unsigned long long)next)[i];
        } else {
            const int32_t *p = next + i;
            int v;
            asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
            i = v;
        }
    }
    if (lane == 0) out[blockIdx.x] = i;
}

template <int MODE>
static void run(const char *name, const int32_t *next, int32_t *out, int n, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(chase<MODE>, dim3(blocks), dim3(64), 0, 0, next, out, n);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chase<MODE>, dim3(blocks), dim3(64), 0, 0, next, out, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s blocks %6d: %8.3f ms  %7.1f ns per dependent read per wave  %.3e reads/s\n", name, blocks, ms,
           ms * 1e6 / STEPS, (double)blocks * STEPS / (ms * 1e-3));
}

int main()
{
    const int n = 1 << 16; // 256 KB: L2-resident
    std::vector<int32_t> h(n);
    uint32_t x = 12345;
    for (int i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (int32_t)(x >> 8) % n; }
    int32_t *next, *out;
    hipMalloc(&next, n * 4); hipMalloc(&out, 1 << 20);
    hipMemcpy(next, h.data(), n * 4, hipMemcpyHostToDevice);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    for (int per_cu : {1, 4, 32, 64}) {
        const int blocks = prop.multiProcessorCount * per_cu;
        run<0>("all lanes, same address", next, out, n, blocks);
        run<1>("lane 0 + readfirstlane", next, out, n, blocks);
        run<2>("s_load (scalar cache)", next, out, n, blocks);
        run<3>("s_load glc", next, out, n, blocks);
        run<4>("vector load, all 64 lanes", next, out, n, blocks);
    }
    return 0;
}
