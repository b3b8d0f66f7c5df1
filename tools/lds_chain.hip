// What does one link of a lone wave's dependent LDS chain cost?  (the rollout walk of uct_lone_kernel: v_and -> v_mad -> ds_read_u16 ->
// s_waitcnt).  One wave per workgroup, `grid` workgroups; per variant the ticks (s_memtime) per link over 4096 links.
//   hipcc --offload-arch=gfx950 -O2 tools/lds_chain.hip -o build_variants/lds_chain && build_variants/lds_chain
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(long long *out, int n)
{
    __shared__ unsigned short tab[32768];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) tab[i] = (unsigned short)((i * 7 + 3) & 0x3fff);
    __syncthreads();
    if (threadIdx.x >= 64) return;
    typedef __attribute__((address_space(3))) const unsigned short lds16;
    typedef __attribute__((address_space(3))) const unsigned lds32;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short *)tab;
    unsigned v = 5, w = 0;
    const long long t0 = clock64();
    if (MODE == 0)       // bare chain: address = value * 2 + base (one v_lshl_add), u16, every lane the same address
        for (int i = 0; i < n; ++i) { unsigned ad = (v << 1) + base; asm volatile("" : "+v"(ad)); v = *(lds16 *)(uintptr_t)ad; }
    if (MODE == 1)       // b32 reads (aligned)
        for (int i = 0; i < n; ++i) { unsigned ad = ((v & 0x1fffu) << 2) + base; asm volatile("" : "+v"(ad)); v = *(lds32 *)(uintptr_t)ad & 0x3fffu; }
    if (MODE == 2)       // the walk's chain: v_and, v_mad_u32_u24, ds_read_u16
        for (int i = 0; i < n; ++i) { unsigned ad = (v & 0x7fffu) * 2u + base; asm volatile("" : "+v"(ad)); v = *(lds16 *)(uintptr_t)ad; }
    if (MODE == 3) {     // bare chain, one active lane
        if (threadIdx.x == 0)
            for (int i = 0; i < n; ++i) { unsigned ad = (v << 1) + base; asm volatile("" : "+v"(ad)); v = *(lds16 *)(uintptr_t)ad; }
    }
    if (MODE == 4)       // bare chain + a readfirstlane and a scalar test per link (the walk's exit test)
        for (int i = 0; i < n; ++i) {
            unsigned ad = (v << 1) + base; asm volatile("" : "+v"(ad)); v = *(lds16 *)(uintptr_t)ad;
            if (__builtin_amdgcn_readfirstlane((int)v) & 0x8000) break;
        }
    if (MODE == 5)       // every lane its own address (no broadcast)
        { v = 5 + threadIdx.x * 64; for (int i = 0; i < n; ++i) { unsigned ad = (v << 1) + base; asm volatile("" : "+v"(ad)); v = *(lds16 *)(uintptr_t)ad; } }
    if (MODE == 6)       // bare chain with 8 independent VALU instructions per link beside it
        for (int i = 0; i < n; ++i) {
            unsigned ad = (v << 1) + base; asm volatile("" : "+v"(ad)); v = *(lds16 *)(uintptr_t)ad;
            asm volatile("v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n"
                         "v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0" : "+v"(w));
        }
    if (MODE == 7)       // ... 16 of them
        for (int i = 0; i < n; ++i) {
            unsigned ad = (v << 1) + base; asm volatile("" : "+v"(ad)); v = *(lds16 *)(uintptr_t)ad;
            asm volatile("v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n"
                         "v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n"
                         "v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n"
                         "v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0" : "+v"(w));
        }
    if (MODE == 8)       // ... 8 scalar instructions
        for (int i = 0; i < n; ++i) {
            unsigned ad = (v << 1) + base; asm volatile("" : "+v"(ad)); v = *(lds16 *)(uintptr_t)ad;
            unsigned sw = 0;
            asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                         "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(sw) : : "scc");
            w += sw;
        }
    if (MODE >= 9 && MODE <= 12) {   // bare chain + NB never-taken conditional branches per link (in the read's shadow), loop unrolled by U
        unsigned sw = 0, t;
        constexpr int NB = MODE == 9 ? 1 : MODE == 10 ? 2 : MODE == 11 ? 4 : 2;
        constexpr int U = MODE == 12 ? 4 : 1;
        for (int i = 0; i < n; i += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                unsigned ad = (v << 1) + base;
                asm volatile("ds_read_u16 %0, %1" : "=v"(t) : "v"(ad) : "memory");
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    asm volatile("s_cmp_eq_u32 %0, 77\n s_cbranch_scc1 9f\n s_add_u32 %0, %0, 0\n 9:" : "+s"(sw) : : "scc");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t) : : "memory");
                v = t;
            }
        }
        w += sw;
    }
    if (MODE == 13) {   // the walk's half-step as written in uct_lone_kernel (tests never true)
        unsigned t, v2 = 0, sv = 0, e = 0, h = 1, h1 = 0, a2 = base, hend = 1u << 30, lane = threadIdx.x, act2 = base;
        asm volatile("1:\n"
            "v_and_b32 %[t], 0x7fff, %[v1]\n v_mad_u32_u24 %[t], %[t], 2, %[a2]\n ds_read_u16 %[v2], %[t]\n v_cmp_eq_u32 vcc, %[h], %[lane]\n"
            "v_readfirstlane_b32 %[e], %[v1]\n s_add_i32 %[h1], %[h], 1\n v_cndmask_b32 %[sv], %[sv], %[v1], vcc\n v_readlane_b32 %[a2], %[act2], 0\n"
            "s_bitcmp1_b32 %[e], 15\n s_cbranch_scc1 2f\n s_cmp_ge_u32 %[h], %[hend]\n s_cbranch_scc1 2f\n s_mov_b32 %[h], %[h1]\n s_waitcnt lgkmcnt(0)\n"
            "v_and_b32 %[t], 0x7fff, %[v2]\n v_mad_u32_u24 %[t], %[t], 2, %[a2]\n ds_read_u16 %[v1], %[t]\n v_cmp_eq_u32 vcc, %[h], %[lane]\n"
            "v_readfirstlane_b32 %[e], %[v2]\n s_add_i32 %[h1], %[h], 1\n v_cndmask_b32 %[sv], %[sv], %[v2], vcc\n v_readlane_b32 %[a2], %[act2], 0\n"
            "s_bitcmp1_b32 %[e], 15\n s_cbranch_scc1 2f\n s_cmp_ge_u32 %[h], %[hend]\n s_cbranch_scc1 2f\n s_mov_b32 %[h], %[h1]\n s_waitcnt lgkmcnt(0)\n"
            "s_cmp_lt_u32 %[h], %[n]\n s_cbranch_scc1 1b\n 2:\n s_waitcnt lgkmcnt(0)"
            : [v1] "+v"(v), [v2] "+v"(v2), [sv] "+v"(sv), [t] "=&v"(t), [e] "+s"(e), [h] "+s"(h), [h1] "+s"(h1), [a2] "+s"(a2)
            : [lane] "v"(lane), [act2] "v"(act2), [hend] "s"(hend), [n] "s"(n) : "vcc", "scc", "memory");
        w += sv + e;
    }
    if (MODE >= 14 && MODE <= 19) {   // the branch-free slot of the grouped walk, with parts left out
        unsigned t, w1, w2, w3, w4, sv = 0, e = 0, h = 1, a2 = base, lane = threadIdx.x, act2 = base, acc = 0, cnt = n / 4;
#define NOTE(WP) "v_cmp_eq_u32 vcc, %[h], %[lane]\n v_nop\n v_nop\n v_cndmask_b32 %[sv], %[sv], " WP ", vcc\n"
#define RFL(WP) "v_readfirstlane_b32 %[e], " WP "\n s_nop 0\n s_lshr_b32 %[e], %[e], 15\n s_lshl1_add_u32 %[acc], %[acc], %[e]\n"
#define RDL "s_add_i32 %[h], %[h], 1\n v_readlane_b32 %[a2], %[act2], 0\n"
#define SLOT(WP, WN, X) "v_and_b32 %[t], 0x7fff, " WP "\n v_mad_u32_u24 %[t], %[t], 2, %[a2]\n ds_read_u16 " WN ", %[t]\n" X "s_waitcnt lgkmcnt(0)\n"
#define GROUP(X1, X2, X3, X4) asm volatile("1:\n" SLOT("%[v1]", "%[w1]", X1) SLOT("%[w1]", "%[w2]", X2) SLOT("%[w2]", "%[w3]", X3) SLOT("%[w3]", "%[w4]", X4) \
            "v_mov_b32 %[v1], %[w4]\n s_sub_u32 %[cnt], %[cnt], 1\n s_cmp_lg_u32 %[cnt], 0\n s_cbranch_scc1 1b\n"                                   \
            : [v1] "+v"(v), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [w4] "=&v"(w4), [sv] "+v"(sv), [t] "=&v"(t), [e] "+s"(e), [h] "+s"(h),         \
              [a2] "+s"(a2), [acc] "+s"(acc), [cnt] "+s"(cnt) : [lane] "v"(lane), [act2] "v"(act2) : "vcc", "scc", "memory")
        if (MODE == 14) GROUP("", "", "", "");
        if (MODE == 15) GROUP(NOTE("%[v1]"), NOTE("%[w1]"), NOTE("%[w2]"), NOTE("%[w3]"));
        if (MODE == 16) GROUP(RFL("%[v1]"), RFL("%[w1]"), RFL("%[w2]"), RFL("%[w3]"));
        if (MODE == 17) GROUP(RDL, RDL, RDL, RDL);
        if (MODE == 18) GROUP(NOTE("%[v1]") RFL("%[v1]") RDL, NOTE("%[w1]") RFL("%[w1]") RDL, NOTE("%[w2]") RFL("%[w2]") RDL, NOTE("%[w3]") RFL("%[w3]") RDL);
        if (MODE == 19) GROUP(NOTE("%[v1]") RFL("%[v1]"), NOTE("%[w1]") RFL("%[w1]"), NOTE("%[w2]") RFL("%[w2]"), NOTE("%[w3]") RFL("%[w3]"));
        w += sv + e + acc;
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (v + w == 0x12345678u) out[1] = v;
}

int main()
{
    long long *d;
    hipMalloc(&d, 16);
    const char *names[20] = {"u16, broadcast address", "b32, broadcast address", "v_and + v_mad + u16 (the walk)", "u16, one active lane",
                            "u16 + readfirstlane + scalar test", "u16, 64 addresses", "u16 + 8 VALU beside", "u16 + 16 VALU beside", "u16 + 8 SALU beside",
                            "u16 + 1 untaken branch in the shadow", "u16 + 2 untaken branches", "u16 + 4 untaken branches", "u16 + 2 untaken, loop unrolled x4",
                            "the walk's half-step (asm of the kernel)",
                            "group of 4 slots: chain only", "  + note (v_cmp, v_cndmask)", "  + readfirstlane, s_lshr, s_lshl1_add", "  + s_add, v_readlane (action term)",
                            "  + all three", "  + note + readfirstlane part"};
    const int n = 4096;
    for (int grid : {1})
        for (int threads : {64})
            for (int mode = 0; mode < 20; ++mode) {
                for (int rep = 0; rep < 2; ++rep) {
    #define L(M) if (mode == M) hipLaunchKernelGGL(k<M>, dim3(grid), dim3(threads), 0, 0, d, n)
                    L(0); L(1); L(2); L(3); L(4); L(5); L(6); L(7); L(8); L(9); L(10); L(11); L(12); L(13); L(14); L(15); L(16); L(17); L(18); L(19);
                    hipDeviceSynchronize();
                }
                long long h[2];
                hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                printf("grid %3d threads %4d  %-36s %.1f ticks per link\n", grid, threads, names[mode], (double)h[0] / n);
            }
    return 0;
}
