// gather_calib.hip -- what do rocprofv3's FETCH_SIZE / WRITE_SIZE report for the ACCESS PATTERNS of the tree-search
// kernels?  (MI355X_MICROARCH.md §HBM calibrates FETCH_SIZE on wide coalesced streams only: "other access widths and
// WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".)
//
// Every kernel below makes a KNOWN number of requests to a KNOWN set of distinct 128-byte lines of a table far larger
// than L2 + Infinity Cache (default 4 GiB), each line touched exactly once per launch, so that
//     counter value / requests  =  bytes the counter tallies per scattered request of that kind,
// and wall time / requests bounds what the fabric really moved (requests/s x 128 B above the measured stream
// ceiling would mean the fabric moves less than a line per request).
//
//   cal_stream_rd16    control: coalesced 16 B per lane over the whole table          (known bytes = table)
//   cal_stream_wr16    control: coalesced 16 B stores over the whole table
//   cal_gather16       one 16 B load per lane at a scattered line (the UCT / OPD node and model-record gathers)
//   cal_gather16_pair  two 16 B loads per lane in the SAME 128 B line, different 64 B halves
//   cal_gather16_x4    four 16 B loads per lane covering 64 contiguous bytes of one line (children block of a node)
//   cal_rmw16          16 B load + 16 B store to the same scattered record (backup read-modify-write)
//   cal_scatter16      16 B store to a scattered record, no load (expansion writes)
//   cal_scatter12      8 B + 4 B stores to a scattered record (value f64 + count i32: the retained-statistics backup)
//
// Lane -> line map: line = (i * ODD) mod n_lines with n_lines a power of two: a bijection, neighbouring lanes land
// 0x9E3779B1 lines apart (no two lanes of a wave share a line, nothing is coalesced).
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_calib.hip -o build_variants/gather_calib
// Run  : build_variants/gather_calib [log2_lines=25]          (prints requests, ms, requests/s per kernel)
//        rocprofv3 --kernel-trace --pmc FETCH_SIZE -- build_variants/gather_calib ;  same with WRITE_SIZE
//        python tools/summarize_calib.py <dir>  ->  profiles/r02_gather_calib.{md,json}
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                                        \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); }   \
    } while (0)

constexpr uint32_t kOdd = 0x9E3779B1u;

__device__ __forceinline__ size_t line_of(size_t i, uint32_t mask) { return (size_t)(((uint32_t)i * kOdd) & mask); }

__global__ void cal_stream_rd16(const uint4 *__restrict__ t, size_t n16, uint32_t *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n16; i += stride) { const uint4 v = t[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void cal_stream_wr16(uint4 *__restrict__ t, size_t n16)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) t[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

__global__ void cal_gather16(const uint4 *__restrict__ t, size_t n_req, uint32_t mask, uint32_t *sink)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_req) return;
    const size_t line = line_of(i, mask);
    const uint4 v = t[line * 8 + (i & 7)];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[0] = v.x;
}

__global__ void cal_gather16_pair(const uint4 *__restrict__ t, size_t n_req, uint32_t mask, uint32_t *sink)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_req) return;
    const size_t line = line_of(i, mask);
    const uint4 a = t[line * 8 + (i & 3)], b = t[line * 8 + 4 + (i & 3)];
    if ((a.x ^ b.y) == 0x12345678u) sink[0] = a.x;
}

__global__ void cal_gather16_x4(const uint4 *__restrict__ t, size_t n_req, uint32_t mask, uint32_t *sink)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_req) return;
    const size_t line = line_of(i, mask);
    const uint4 *p = t + line * 8 + (i & 1) * 4;
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    if ((a.x ^ b.y ^ c.z ^ d.w) == 0x12345678u) sink[0] = a.x;
}

__global__ void cal_rmw16(uint4 *__restrict__ t, size_t n_req, uint32_t mask)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_req) return;
    const size_t line = line_of(i, mask);
    uint4 v = t[line * 8 + (i & 7)];
    v.x += 1u; v.z ^= (uint32_t)i;
    t[line * 8 + (i & 7)] = v;
}

__global__ void cal_scatter16(uint4 *__restrict__ t, size_t n_req, uint32_t mask)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_req) return;
    const size_t line = line_of(i, mask);
    t[line * 8 + (i & 7)] = make_uint4((uint32_t)i, 0u, 0xffffffffu, 7u);
}

__global__ void cal_scatter12(uint4 *__restrict__ t, size_t n_req, uint32_t mask)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_req) return;
    const size_t line = line_of(i, mask);
    uint4 *rec = t + line * 8 + (i & 7);
    *reinterpret_cast<double *>(rec) = (double)i;                 // value f64
    reinterpret_cast<int32_t *>(rec)[2] = (int32_t)i;             // count i32 (first_child left alone)
}

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 25;
    if (lg < 10 || lg > 27) { fprintf(stderr, "log2_lines must be in [10, 27]\n"); return 1; }
    const size_t n_lines = (size_t)1 << lg, bytes = n_lines * 128, n16 = bytes / 16;
    const uint32_t mask = (uint32_t)(n_lines - 1);
    uint4 *t;
    uint32_t *sink;
    CK(hipMalloc(&t, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(t, 1, bytes));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("table %.2f GiB = %zu lines of 128 B; one request per line and launch\n", bytes / 1073741824.0, n_lines);
    const size_t n_req = n_lines;
    const dim3 blk(256), grd((unsigned)((n_req + 255) / 256)), sgrd(8192);
    auto timed = [&](const char *name, auto launch, double useful_bytes_per_req, size_t reqs) {
        launch();                                   // warm-up (also the launch a PMC pass averages with the next two)
        CK(hipEventRecord(e0));
        for (int r = 0; r < 2; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 2;
        printf("%-18s requests %zu  useful B/request %.0f  %.3f ms  %.2f G requests/s  useful %.3f TB/s  "
               "(x64 B = %.2f TB/s, x128 B = %.2f TB/s)\n", name, reqs, useful_bytes_per_req, ms, reqs / (ms * 1e6),
               reqs * useful_bytes_per_req / (ms * 1e9), reqs * 64.0 / (ms * 1e9), reqs * 128.0 / (ms * 1e9));
    };
    timed("cal_stream_rd16", [&] { hipLaunchKernelGGL(cal_stream_rd16, sgrd, blk, 0, 0, t, n16, sink); }, 16, n16);
    timed("cal_stream_wr16", [&] { hipLaunchKernelGGL(cal_stream_wr16, sgrd, blk, 0, 0, t, n16); }, 16, n16);
    timed("cal_gather16", [&] { hipLaunchKernelGGL(cal_gather16, grd, blk, 0, 0, t, n_req, mask, sink); }, 16, n_req);
    timed("cal_gather16_pair", [&] { hipLaunchKernelGGL(cal_gather16_pair, grd, blk, 0, 0, t, n_req, mask, sink); }, 32, n_req);
    timed("cal_gather16_x4", [&] { hipLaunchKernelGGL(cal_gather16_x4, grd, blk, 0, 0, t, n_req, mask, sink); }, 64, n_req);
    timed("cal_rmw16", [&] { hipLaunchKernelGGL(cal_rmw16, grd, blk, 0, 0, t, n_req, mask); }, 32, n_req);
    timed("cal_scatter16", [&] { hipLaunchKernelGGL(cal_scatter16, grd, blk, 0, 0, t, n_req, mask); }, 16, n_req);
    timed("cal_scatter12", [&] { hipLaunchKernelGGL(cal_scatter12, grd, blk, 0, 0, t, n_req, mask); }, 12, n_req);
    CK(hipDeviceSynchronize());
    return 0;
}
