// Where do the wavefronts of a launch land?  G workgroups of W waves, each wave busy for a while; every wave records its
// HW_ID (SIMD, CU, SH, SE) and XCC_ID.  Prints the histogram of waves per SIMD / per CU.
//   hipcc --offload-arch=gfx950 -O2 tools/wave_placement.hip -o /tmp/wave_placement && /tmp/wave_placement G W [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void k(unsigned *out, int spin)
{
    extern __shared__ char lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    double x = threadIdx.x * 1e-3;
    for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = hw;
        out[2 * w + 1] = xcc | (x > 1e30 ? 1u << 31 : 0u);
    }
    if (x == 12345.678) lds[threadIdx.x] = 1;
}

int main(int argc, char **argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 1024, W = argc > 2 ? atoi(argv[2]) : 1, L = argc > 3 ? atoi(argv[3]) : 0;
    unsigned *d;
    hipMalloc(&d, (size_t)G * W * 8);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, L);
    hipLaunchKernelGGL(k, dim3(G), dim3(64 * W), L, 0, d, 200000);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(G), dim3(64 * W), L, 0, d, 200000);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h((size_t)G * W * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_simd, per_cu;
    for (int w = 0; w < G * W; ++w) {
        const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned cu_key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        per_cu[cu_key]++;
        per_simd[(cu_key << 2) | simd]++;
    }
    std::map<int, int> hist_simd, hist_cu;
    for (auto &p : per_simd) hist_simd[p.second]++;
    for (auto &p : per_cu) hist_cu[p.second]++;
    printf("G=%d workgroups x %d waves, LDS %d B: %.3f ms; CUs used %zu, SIMDs used %zu\n", G, W, L, ms, per_cu.size(), per_simd.size());
    printf("  waves per CU  :"); for (auto &p : hist_cu) printf(" %d CUs x %d", p.second, p.first); printf("\n");
    printf("  waves per SIMD:"); for (auto &p : hist_simd) printf(" %d SIMDs x %d", p.second, p.first); printf("\n");
    return 0;
}
