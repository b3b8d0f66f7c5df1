// opd_closing.hpp -- the closing passes of the LDS-resident optimistic planners (opd_kernel, ropd_kernel): every
// backup_to_root of the plan at once (deterministic.py:67-79) and get_plan (abstract.py:143-156), on the EXPANSION tree.
//
// After K expansions the final lower bound of an expanded node is the maximum of the creation-time lower bounds of the
// LEAVES below it.  With s_k = max over the leaf children of expansion k and up(k) = the expansion that created k's node,
//     L_k = max(s_k, max over {j : up(j) = k} of L_j)
// is a subtree maximum, computed here by pointer jumping: in round r every expansion pushes what it has gathered so far
// to its 2^(r-1)-th ancestor (an LDS ds_max_f64) and doubles its link; ceil(log2(depth + 1)) rounds, each one pass of
// one expansion per lane.  The bench trees are deep (BASELINE C4: ~160 levels for 1 000 expansions, long runs where
// expansion k+1 takes a child of expansion k), so the reference's order -- a K-long chain of read / max / write -- or
// any level-by-level schedule is a chain of hundreds of dependent steps; pointer jumping needs eight.  max is exact and
// idempotent, so neither the order nor a contribution arriving twice changes a bit of the result.
//
// The plan descent is then a walk over a per-expansion table {action, next expansion} prepared for all expansions at once;
// only a level with TIED maxima (where the reference draws from the generator) is evaluated during the walk.
//
// LDS use (the bounds array of the main loop is dead by now): val f64[K] | link i32[K] | link' i32[K] | node->k i32[cap].
#pragma once
#include "pcg64.hpp"
#include "wave.hpp"

namespace mp {

__host__ __device__ inline bool closing_compact_fits(long K, long A, long cap, long lds_bytes)
{
    return A <= 255 && K < (1 << 22) && 16 * K + 4 * cap <= lds_bytes;
}

// loadL(id): creation-time lower bound of node id (global memory); storeL(id, v): final lower bound of an expanded node.
// Returns the plan length; plan_row may be null.  One wavefront; all 64 lanes call.
template <class LoadL, class StoreL>
__device__ __forceinline__ int closing_compact(void *lds, int K, int k_done, int n_nodes, int A, const int32_t *exp_map,
                                               LoadL loadL, StoreL storeL, Pcg64 &gen, int32_t *plan_row, int max_plan_len,
                                               double &root_lower)
{
    const int lane = threadIdx.x;
    const double ninf = -INFINITY;
    double *val = reinterpret_cast<double *>(lds);
    int32_t *link0 = reinterpret_cast<int32_t *>(val + K), *link1 = link0 + K, *nodek = link1 + K;

#ifdef MP_PROFILE
    long long tc[8]; int n_rounds = 0, n_tied = 0;
#define CT(i) tc[i] = clock64()
#else
#define CT(i)
#endif
    CT(0);
    for (int i = lane; i < n_nodes; i += 64) nodek[i] = -1;
    __builtin_amdgcn_wave_barrier(); // (one wavefront: its LDS operations execute in program order)
    for (int k = lane; k < k_done; k += 64) {
        const int n = exp_map[k];
        nodek[n] = k;
        link0[k] = n > 0 ? (n - 1) / A : -1; // node n is child (n - 1) % A of expansion (n - 1) / A
    }
    __builtin_amdgcn_wave_barrier();
    CT(1);
    // s_k: the leaf children's creation-time bounds (all |A| records requested, eight in flight; an expanded child's is dropped)
    for (int kb = 0; kb < k_done; kb += 64) {
        const int k = kb + lane;
        if (k < k_done) {
            const int g = 1 + k * A;
            double m = ninf;
            for (int a0 = 0; a0 < A; a0 += 8) {
                double l[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) l[j] = a0 + j < A ? loadL(g + a0 + j) : ninf;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool leaf = a0 + j < A && nodek[g + (a0 + j < A ? a0 + j : 0)] < 0;
                    if (leaf && l[j] > m) m = l[j];
                }
            }
            val[k] = m;
        }
    }
    __builtin_amdgcn_wave_barrier();
    CT(2);
    // pointer jumping
    int32_t *cur = link0, *nxt = link1;
    for (;;) {
        bool active = false;
        for (int kb = 0; kb < k_done; kb += 256) { // four chunks at a time: their dependent LDS round trips overlap
            int u[4], un[4];
            double v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kb + 64 * j + lane;
                u[j] = k < k_done ? cur[k] : -1;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kb + 64 * j + lane;
                v[j] = u[j] >= 0 ? val[k] : ninf;
                un[j] = u[j] >= 0 ? cur[u[j]] : -1;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kb + 64 * j + lane;
                if (u[j] >= 0) {
                    __hip_atomic_fetch_max(&val[u[j]], v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    active = true;
                }
                if (k < k_done) nxt[k] = un[j];
            }
        }
        __builtin_amdgcn_wave_barrier();
        int32_t *t = cur; cur = nxt; nxt = t;
#ifdef MP_PROFILE
        ++n_rounds;
#endif
        if (!any64(active)) break;
    }
    CT(3);
    // the final bounds into the records (the tree export reads them); the root's for the caller
    for (int k = lane; k < k_done; k += 64) storeL(exp_map[k], val[k]);
    root_lower = k_done > 0 ? val[0] : loadL(0);

    CT(4);
    // per expansion: the child of the plan when the maximum is unique -- {next expansion + 1, action}, 0 = the child is a
    // leaf (the plan ends) -- or -1: tied maxima, resolved during the walk with the generator (deterministic.py:21-26)
    int32_t *step = nxt;
    for (int kb = 0; kb < k_done; kb += 64) {
        const int k = kb + lane;
        if (k < k_done) {
            const int g = 1 + k * A;
            double m = ninf;
            int nt = 0, a_first = 0, k_first = nodek[g];
            for (int a0 = 0; a0 < A; a0 += 8) {
                double l[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) l[j] = a0 + j < A ? loadL(g + a0 + j) : ninf;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (a0 + j < A) {
                        const int kc = nodek[g + a0 + j];
                        const double v = kc >= 0 ? val[kc] : l[j];
                        if (v > m) { m = v; nt = 1; a_first = a0 + j; k_first = kc; }
                        else if (v == m) ++nt;
                    }
            }
            step[k] = nt == 1 ? (((k_first + 1) << 8) | a_first) : -1;
        }
    }
    __builtin_amdgcn_wave_barrier();
    CT(5);
    int len = 0;
    int kcur = k_done > 0 ? 0 : -1; // the first expansion is always the root
    while (kcur >= 0) {
        const int s = __builtin_amdgcn_readfirstlane(step[kcur]);
        int a, knext;
        if (s != -1) {
            a = s & 255;
            knext = (s >> 8) - 1;
        } else {
#ifdef MP_PROFILE
            ++n_tied;
#endif
            const int fc = 1 + kcur * A;
            int kc = -1;
            double l = ninf;
            if (lane < A) {
                kc = nodek[fc + lane];
                l = kc >= 0 ? val[kc] : loadL(fc + lane);
            }
            const double m = A <= 16 ? row0_max(l) : wave_max(l);
            const unsigned long long ties = ballot64(lane < A && l == m);
            const int nt = __popcll(ties);
            int pick = (int)gen.below((uint32_t)nt); // uniform across lanes (same state, same draws)
            unsigned long long t = ties;
            while (pick-- > 0) t &= t - 1;
            a = __ffsll((long long)t) - 1;
            knext = __builtin_amdgcn_readlane(kc, a);
        }
        if (lane == 0 && plan_row && len < max_plan_len) plan_row[len] = a;
        ++len;
        kcur = knext;
    }
    CT(6);
#ifdef MP_PROFILE
    if (blockIdx.x == 0 && lane == 0)
        printf("closing prof root0: tables %lld  s_k %lld  jumping %lld (%d rounds)  store %lld  steps %lld  walk %lld (%d levels, %d tied)\n",
               tc[1] - tc[0], tc[2] - tc[1], tc[3] - tc[2], n_rounds, tc[4] - tc[3], tc[5] - tc[4], tc[6] - tc[5], len, n_tied);
#endif
#undef CT
    return len;
}

} // namespace mp
