// Read-only HBM streaming ceiling on this GPU: sum 4 GB of doubles with 16-byte (and 32-byte) per-lane loads.
// Build: hipcc --offload-arch=gfx950 -O3 tools/stream_read.hip -o build_variants/stream_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double double2_t __attribute__((ext_vector_type(2)));
__global__ void rd(const double2_t *__restrict__ p, size_t n, double *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const double2_t a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    }
    for (; i < n; i += stride) acc += p[i].x + p[i].y;
    if (acc == 123.456) out[0] = acc;
}
int main(int argc, char **argv)
{
    const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4000000000ull, n = bytes / 16;
    double2_t *p; double *o;
    hipMalloc(&p, bytes); hipMalloc(&o, 8);
    hipMemset(p, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1024, 2048, 4096, 8192}) {
        hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, 0, p, n, o);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, 0, p, n, o);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("read-only stream, %d blocks x 256: %.3f ms per %.1f GB -> %.2f TB/s\n", blocks, ms / 5, bytes / 1e9, bytes / (ms / 5 * 1e-3) / 1e12);
    }
    return 0;
}
