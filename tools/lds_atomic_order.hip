// Do LDS atomics of ONE wavefront instruction that hit the same address apply in LANE ORDER on gfx950?
// ds_min_rtn_f64 from 64 lanes onto few addresses with random values: the value a lane gets back must be the minimum of the
// initial value and the values of the LOWER lanes with the same address (prefix minimum in lane order).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
__global__ void k(const double *vals, const int *addr, const unsigned long long *maskp, double *ret, double *fin, int trials)
{
    __shared__ double cell[8];
    const int lane = threadIdx.x;
    for (int t = 0; t < trials; ++t) {
        if (lane < 8) cell[lane] = 5.0;
        __syncthreads();
        const int i = (blockIdx.x * trials + t) * 64 + lane;
        double r = -1.0;
        if ((maskp[blockIdx.x * trials + t] >> lane) & 1ULL) r = __hip_atomic_fetch_min(&cell[addr[i]], vals[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ret[i] = r;
        __syncthreads();
        if (lane < 8) fin[(blockIdx.x * trials + t) * 8 + lane] = cell[lane];
        __syncthreads();
    }
}
int main()
{
    const int blocks = 2048, trials = 64, n = blocks * trials * 64;
    double *vals = (double *)malloc(n * 8), *ret = (double *)malloc(n * 8), *fin = (double *)malloc(blocks * trials * 64);
    int *addr = (int *)malloc(n * 4);
    unsigned long long *mask = (unsigned long long *)malloc(blocks * trials * 8);
    srand(1);
    for (int i = 0; i < n; ++i) { vals[i] = (rand() % 64) / 8.0; addr[i] = rand() % ((i / 64) % 8 + 1); }
    for (int i = 0; i < blocks * trials; ++i) mask[i] = ((unsigned long long)rand() << 40) ^ ((unsigned long long)rand() << 20) ^ rand() ^ ((i & 3) == 0 ? ~0ULL : 0ULL);
    double *dv, *dr, *df; int *da; unsigned long long *dm;
    hipMalloc(&dv, n * 8); hipMalloc(&dr, n * 8); hipMalloc(&df, blocks * trials * 64); hipMalloc(&da, n * 4); hipMalloc(&dm, blocks * trials * 8);
    hipMemcpy(dv, vals, n * 8, hipMemcpyHostToDevice); hipMemcpy(da, addr, n * 4, hipMemcpyHostToDevice); hipMemcpy(dm, mask, blocks * trials * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, dv, da, dm, dr, df, trials);
    hipMemcpy(ret, dr, n * 8, hipMemcpyDeviceToHost); hipMemcpy(fin, df, blocks * trials * 64, hipMemcpyDeviceToHost);
    long bad = 0, badfin = 0;
    for (int w = 0; w < blocks * trials; ++w) {
        double cell[8]; for (int c = 0; c < 8; ++c) cell[c] = 5.0;
        for (int l = 0; l < 64; ++l) {
            const int i = w * 64 + l;
            if (!((mask[w] >> l) & 1ULL)) continue;
            if (ret[i] != cell[addr[i]]) ++bad;
            cell[addr[i]] = fmin(cell[addr[i]], vals[i]);
        }
        for (int c = 0; c < 8; ++c) badfin += fin[w * 8 + c] != cell[c];
    }
    printf("ds_min_rtn_f64 lane-order check: %d wave instructions, %ld returned values out of lane order, %ld wrong final cells\n", blocks * trials, bad, badfin);
    return bad || badfin;
}
