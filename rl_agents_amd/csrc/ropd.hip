// ropd.hip -- discrete robust optimistic planning (agents/robust/robust.py:28-50 over tree_search/deterministic.py).
//
// DiscreteRobustPlanner plans on a JOINT environment of M models stepped together (robust.py:9-16): a joint state is M
// state indices, a step yields M rewards and M terminal flags, so DeterministicNode.update takes its ndarray branch
// (deterministic.py:54-59) and a leaf's lower / upper bounds are VECTORS over the models, which RobustNode reads
// through np.min (robust.py:42-49):
//   * leaf to expand   = first maximal  min_m U[m]   in leaves order            (robust.py:37)
//   * children         L_c[m] = L_leaf[m] + gamma^(d-1) r_m,  U_c[m] = L_c[m] + gamma^d / (1 - gamma),
//                      terminal models: L_c[m] = U_c[m] = L_c[m] + terminal_reward gamma^d / (1 - gamma)
//   * backup_to_root   an expanded node's bounds become the SCALARS max_c min_m L_c, max_c min_m U_c
//                      (deterministic.py:74-79 with RobustNode.get_value_*_bound)
//   * plan             children with maximal min_m L, random ties                (deterministic.py:21-26)
//
// Mapping = opd.hip's: ONE ROOT PER WAVEFRONT; the scalar key min_m U of every node lives in the class-contiguous
// upper-bound array (LDS, or HBM/L2 for big batches) with the two-level cached argmax; lane a creates child a, looping
// over the M models (one 16-byte model-record gather per model, all M in flight together); the scalar backups are
// deferred to one bottom-up pass at the end exactly as in opd.hip -- no planning decision reads an expanded node's
// bounds, children start from their parent's CREATION-TIME vector, and max is exact.
// HBM per node: M x {L f64, state i32, reward f64} + {min_m L, min_m U, depth, done bits}.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.hpp"
#include "pcg64.hpp"
#include "wave.hpp"
#include "opd_closing.hpp"

namespace mp {

// done flags of the models beyond a node's 32 flag bits ride in the stored reward: BIT 62 of its pattern.  A reward the
// planner accepts lies in [0, 1] (deterministic.py:46-47; anything else ends the plan with MP_ERR_REWARD_RANGE): its
// exponent field is at most 0x3ff, so bit 62 is free -- and the sign bit stays the reward's own (-0.0 passes the range check
// and must neither read as "done" nor lose its sign in the export: ADVICE r4).
__host__ __device__ __forceinline__ double reward_with_done(double r)
{
    unsigned long long b;
    memcpy(&b, &r, 8);
    b |= 1ull << 62;
    memcpy(&r, &b, 8);
    return r;
}
__host__ __device__ __forceinline__ bool reward_done(double r)
{
    unsigned long long b;
    memcpy(&b, &r, 8);
    return (b >> 62) & 1ull;
}
__host__ __device__ __forceinline__ double reward_plain(double r)
{
    unsigned long long b;
    memcpy(&b, &r, 8);
    b &= ~(1ull << 62);
    memcpy(&r, &b, 8);
    return r;
}


// (one terminal flag per model in a 32-bit word of the node; from the 33rd model on the flag rides in the sign bit of the
// stored reward -- rewards are range-checked to [0, 1], and 0. with the flag is -0. -- so the number of models is not bounded)

struct ROpdArgs {
    int n_roots, M, S, A, K, cap, done_on_next, max_plan_len;
    int T; // row length of a residue class in the upper-bound array: odd, >= ceil(cap / 64)
    int Tsib, lgP; // ropd_wide_kernel<SIB>: row length of the sibling layout; a leaf's code is (group << lgP) | child (opd.hip)
    int chunk; // ropd_wide_kernel: expansions per LDS window of the closing lower-bound pass (power of two <= 64)
    int closing_chain; // ropd_kernel: 1 = the node-array closing passes even where opd_closing.hpp fits (MP_OPD_CLOSING=chain)
    const Rec *rec;            // [M][S*A] packed records of every model
    const int32_t *root_state; // [n_roots][M]
    const double *g1, *gdiv, *tdiv;
    uint64_t *rng;
    double *Lv;        // [n_roots][cap][M] creation-time lower-bound vectors
    int32_t *Sv;       // [n_roots][cap][M] joint states
    double *Rv;        // [n_roots][cap][M] rewards
    double *Lmin;      // [n_roots][cap]    min_m L at creation; the backed-up scalar for expanded nodes at the end
    double *Umin;      // [n_roots][cap]    min_m U of leaves, -inf for expanded nodes (export fills those in)
    int32_t *meta;     // [n_roots][cap][2] depth, done bits
    double *leaf_global; // [n_roots][64 * T] upper-bound array of the high-occupancy variant (else nullptr)
    int32_t *expanded; // [n_roots][K]
    int32_t *n_nodes_out;
    int32_t *plans, *plan_len, *status;
    double *root_lower, *root_upper;
    int64_t *env_steps;
};

// ropd_kernel<EXPG = false>: upper-bound array and parent map in LDS (44 KB per root at budget 5000: 3 roots per CU);
// ropd_kernel<EXPG = true>: the parent map in HBM (40 448 B: 4 roots per CU, so a 1024-root batch stays on this
// low-latency form); ropd_wide_kernel below: the bounds array in HBM/L2, 8 waves per SIMD (see opd.hip for all three).
template <bool EXPG, int MB>
__global__ __launch_bounds__(64) void ropd_kernel(ROpdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int T = p.T;
    double *leafU = lds;
    int32_t *exp_lds = EXPG ? p.expanded + (long)blockIdx.x * (p.K > 0 ? p.K : 1) : reinterpret_cast<int32_t *>(lds + 64 * T);
#define LU(id) leafU[((id) & 63) * T + ((id) >> 6)]
    const int lane = threadIdx.x, root = blockIdx.x, A = p.A, M = p.M;
    const long base = (long)root * p.cap, SA = (long)p.S * A;
    double *Lv = p.Lv + base * M, *Rv = p.Rv + base * M, *Lmin = p.Lmin + base, *Umin = p.Umin + base;
    int32_t *Sv = p.Sv + base * M, *meta = p.meta + base * 2;
    const uint32_t done_bit = p.done_on_next ? 2u : 1u;
    const double ninf = -INFINITY;

    // deterministic.py:10-19 root: L = U = 0 (scalars), depth 0
    if (lane == 0) {
        for (int m = 0; m < M; ++m) { Lv[m] = 0.0; Sv[m] = p.root_state[(long)root * M + m]; Rv[m] = 0.0; }
        Lmin[0] = 0.0; meta[0] = 0; meta[1] = 0;
        LU(0) = 0.0;
    }
    __syncthreads();
    int n_nodes = 1, status = MP_OK, k_done = 0;
    int real_steps = 0; // children of listed actions = planner.step calls of the joint environment
    double cbu = lane == 0 ? 0.0 : ninf; // best leaf of this lane's class (ids == lane mod 64)
    int cbid = lane == 0 ? 0 : 0x7fffffff;

    if constexpr (MB > 0) {
    // ---- MB > 0: at most MB models, every finite bound >= +0.0 (gamma in [0, 1), terminal reward >= 0).  An expansion of
    // the generic loop below is a chain of 2 M dependent round trips (model m's state of the leaf, then its record, one
    // model after the other).  Here the leaf's M states and bounds are requested right after the selection (they arrive
    // under the class re-scan's LDS reads), the M x |A| model records right after that (they arrive under the re-scan's
    // reduction), and the reductions take the zero-fill DPP steps (wave.hpp).
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    for (int k = 0; k < p.K; ++k) {
        // ---- robust.py:37: first maximal min_m U among the leaves
        double bu = cbu;
        int leaf = cbid;
        wave_argmax_nonneg(bu, leaf);
        const int cls = leaf & 63;
        int sv[MB];
        double lv[MB];
        {
            const double *Lp = Lv + (long)leaf * M;
            const int32_t *Sp = Sv + (long)leaf * M;
#pragma unroll
            for (int m = 0; m < MB; ++m) { // (uniform addresses: broadcast loads; m >= M re-reads model 0, unused)
                sv[m] = Sp[m < M ? m : 0];
                lv[m] = Lp[m < M ? m : 0];
            }
        }
        const int dleaf = meta[2 * leaf];
        if (lane == 0) LU(leaf) = ninf;
        __builtin_amdgcn_wave_barrier();
        u32x4 rr[MB];
        unsigned long long g1_bits, gdiv_bits, tdiv_bits;
        int d;
        {
            const double *row = leafU + cls * T;
            const int cnt = (n_nodes - cls + 63) >> 6;
            double ru = ninf;
            int rid = 0x7fffffff;
            if (T <= 128) { // at most two entries per lane (budget 5000: 79 per class): no loop, both reads unconditional
                const double u0 = row[lane < cnt ? lane : 0], u1 = row[lane + 64 < cnt ? lane + 64 : 0];
                if (lane < cnt && u0 > ru) { ru = u0; rid = cls + (lane << 6); }
                if (lane + 64 < cnt && u1 > ru) { ru = u1; rid = cls + ((lane + 64) << 6); }
            } else {
                for (int t = lane; t < cnt; t += 128) { // two reads in flight per trip
                    const double u0 = row[t];
                    const double u1 = t + 64 < cnt ? row[t + 64] : ninf;
                    if (u0 > ru) { ru = u0; rid = cls + (t << 6); }
                    if (u1 > ru) { ru = u1; rid = cls + ((t + 64) << 6); }
                }
            }
            // (the leaf's states and bounds are first touched HERE: left alone, the compiler reads a state into an SGPR
            // right behind its load -- a wait for the round trip before the re-scan -- and meets the last bound among the
            // children's stores, where a wait is a wait for those stores)
            int dl = dleaf;
#pragma unroll
            for (int m = 0; m < MB; ++m) asm volatile("" : "+v"(sv[m]), "+v"(lv[m]), "+v"(dl));
            const int cjl = (lane - n_nodes) & 63; // child j is computed by lane (g + j) mod 64, the owner of its class
            // the model records (every lane loads -- the lanes without a child the last action's record, unused) and the gamma-table
            // entries.  Inline assembly: the compiler sinks a plain load to its first use, below the reduction; `ru` / `rid`
            // pass through so that the reduction cannot be scheduled above the requests.  Waited for by hand below.
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const Rec *src = p.rec + ((long)(m < M ? m : 0) * SA + (long)sv[m] * A + (cjl < A ? cjl : A - 1));
                asm volatile("global_load_dwordx4 %0, %2, off" : "=v"(rr[m]), "+v"(ru) : "v"(src) : "memory");
            }
            d = __builtin_amdgcn_readfirstlane(dl) + 1;
            {
                const double *s0 = p.g1 + d, *s1 = p.gdiv + d, *s2 = p.tdiv + d;
                asm volatile("s_load_dwordx2 %0, %4, 0x0\n\ts_load_dwordx2 %1, %5, 0x0\n\ts_load_dwordx2 %2, %6, 0x0"
                             : "=&s"(g1_bits), "=&s"(gdiv_bits), "=&s"(tdiv_bits), "+v"(rid) : "s"(s0), "s"(s1), "s"(s2) : "memory");
            }
            wave_argmax_nonneg(ru, rid);
            if (lane == cls) { cbu = ru; cbid = rid; }
        }
        // ---- DeterministicNode.expand (deterministic.py:28-43), update() with ndarray reward / done (:45-65)
        // (`cbid` passes through: the wait stays below the reduction)
        if (MB == 2)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(rr[0]), "+v"(rr[MB > 1 ? 1 : 0]), "+s"(g1_bits), "+s"(gdiv_bits), "+s"(tdiv_bits), "+v"(cbid) : : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(rr[0]), "+v"(rr[MB > 1 ? 1 : 0]), "+v"(rr[MB > 2 ? 2 : 0]), "+v"(rr[MB > 3 ? 3 : 0]),
                         "+s"(g1_bits), "+s"(gdiv_bits), "+s"(tdiv_bits), "+v"(cbid) : : "memory");
        const double g1d = __longlong_as_double((long long)g1_bits), gdivd = __longlong_as_double((long long)gdiv_bits),
                     tdivd = __longlong_as_double((long long)tdiv_bits);
        const int g = n_nodes;
        bool bad = false, avail = false;
        double Uc_mine = ninf;
        const int cj = (lane - g) & 63;
        if (cj < A) {
            const int c = g + cj;
            double lmin = 0.0, umin = 0.0;
            uint32_t dbits = 0;
#pragma unroll
            for (int m = 0; m < MB; ++m) // JointEnv.step: every model steps its own state (robust.py:13-16)
                if (m < M) {
                    const int32_t nxt = (int32_t)rr[m].x;
                    const uint32_t flags = rr[m].y;
                    const double r = __hiloint2double((int)rr[m].w, (int)rr[m].z);
                    avail |= (flags & 4u) != 0;         // robust.py:22-25: the union of the models' listed actions
                    bad |= !(0.0 <= r) || !(r <= 1.0);  // np.all(0 <= reward), np.all(reward <= 1)
                    const bool dn = (flags & done_bit) != 0;
                    double Lc = lv[m] + g1d * r;
                    double Uc = Lc + gdivd;
                    if (dn) {
                        const double nv = Lc + tdivd;
                        Lc = nv; Uc = nv;
                    }
                    Lv[(long)c * M + m] = Lc;
                    Sv[(long)c * M + m] = nxt;
                    Rv[(long)c * M + m] = (m >= 32 && dn) ? reward_with_done(r) : r; // (models beyond the 32 done bits of the node: the flag rides in bit 62 of the reward)
                    if (m < 32) dbits |= (dn ? 1u : 0u) << m;
                    if (m == 0 || Lc < lmin) lmin = Lc; // np.min
                    if (m == 0 || Uc < umin) umin = Uc;
                }
            bad = bad && avail;                          // (phantom slots: see the generic loop)
            if (!avail) { lmin = ninf; umin = ninf; }
            Lmin[c] = lmin;
            meta[2 * c] = d; meta[2 * c + 1] = (int32_t)dbits;
            LU(c) = umin;
            Uc_mine = umin;
        }
        real_steps += __popcll(ballot64(avail));
        if (lane == 0) exp_lds[k] = leaf;
        n_nodes += A;
        k_done = k + 1;
        if (any64(bad)) { status = MP_ERR_REWARD_RANGE; break; }
        // The next expansion may read these children's vectors through global memory, from other lanes OF THIS WAVEFRONT
        // (the workgroup is one wave): its vector-memory operations are performed in order, so a wavefront-scope fence is
        // all the ordering that needs -- __syncthreads() also waited for the stores' acknowledgements, every expansion.
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (Uc_mine > cbu) { cbu = Uc_mine; cbid = g + cj; } // (-inf in the lanes without a child; on equality the older leaf stays)
    }
    } else {
    for (int k = 0; k < p.K; ++k) {
        // ---- robust.py:37: first maximal min_m U among the leaves
        double bu = cbu;
        int leaf = cbid;
        wave_argmax(bu, leaf);
        const int cls = leaf & 63;
        const int dleaf = meta[2 * leaf]; // (uniform address: one broadcast load, in flight under the class re-scan)
        if (lane == 0) LU(leaf) = ninf;
        __builtin_amdgcn_wave_barrier();
        {
            const double *row = leafU + cls * T;
            const int cnt = (n_nodes - cls + 63) >> 6;
            double ru = ninf;
            int rid = 0x7fffffff;
            if (T <= 128) { // at most two entries per lane (budget 5000: 79 per class): no loop, both reads unconditional
                const double u0 = row[lane < cnt ? lane : 0], u1 = row[lane + 64 < cnt ? lane + 64 : 0];
                if (lane < cnt && u0 > ru) { ru = u0; rid = cls + (lane << 6); }
                if (lane + 64 < cnt && u1 > ru) { ru = u1; rid = cls + ((lane + 64) << 6); }
            } else {
                for (int t = lane; t < cnt; t += 128) { // two reads in flight per trip
                    const double u0 = row[t];
                    const double u1 = t + 64 < cnt ? row[t + 64] : ninf;
                    if (u0 > ru) { ru = u0; rid = cls + (t << 6); }
                    if (u1 > ru) { ru = u1; rid = cls + ((t + 64) << 6); }
                }
            }
            wave_argmax(ru, rid);
            if (lane == cls) { cbu = ru; cbid = rid; }
        }
        // ---- DeterministicNode.expand (deterministic.py:28-43), update() with ndarray reward / done (:45-65)
        const int d = __builtin_amdgcn_readfirstlane(dleaf) + 1;
        typedef const double __attribute__((address_space(4))) *scalar_f64;
        const double g1d = ((scalar_f64)(unsigned long long)p.g1)[d], gdivd = ((scalar_f64)(unsigned long long)p.gdiv)[d],
                     tdivd = ((scalar_f64)(unsigned long long)p.tdiv)[d];
        const int g = n_nodes;
        bool bad = false, avail = false;
        double Uc_mine = 0.0;
        if (lane < A) {
            const int c = g + lane;
            double lmin = 0.0, umin = 0.0;
            uint32_t dbits = 0;
            const double *Lp = Lv + (long)leaf * M;
            const int32_t *Sp = Sv + (long)leaf * M;
            for (int m = 0; m < M; ++m) { // JointEnv.step: every model steps its own state (robust.py:13-16)
                const Rec rc = p.rec[(long)m * SA + (long)Sp[m] * A + lane];
                const double r = rc.reward;
                // robust.py:22-25: the joint env lists the UNION of the actions its models list in their own states
                avail |= (rc.flags & 4u) != 0;
                bad |= !(0.0 <= r) || !(r <= 1.0); // np.all(0 <= reward), np.all(reward <= 1)
                const bool dn = (rc.flags & done_bit) != 0;
                double Lc = Lp[m] + g1d * r;
                double Uc = Lc + gdivd;
                if (dn) {
                    const double nv = Lc + tdivd;
                    Lc = nv; Uc = nv;
                }
                Lv[(long)c * M + m] = Lc;
                Sv[(long)c * M + m] = rc.next;
                Rv[(long)c * M + m] = (m >= 32 && dn) ? reward_with_done(r) : r; // (models beyond the 32 done bits of the node: the flag rides in bit 62 of the reward)
                if (m < 32) dbits |= (dn ? 1u : 0u) << m;
                if (m == 0 || Lc < lmin) lmin = Lc; // np.min
                if (m == 0 || Uc < umin) umin = Uc;
            }
            // deterministic.py:32-35: an action no model lists gets no child; its slot stays in the id space as a PHANTOM
            // with min L = min U = -inf (never the best leaf, never a maximum of the backup, never a tie of the plan;
            // dropped by the export) -- see opd.hip
            bad = bad && avail;
            if (!avail) { lmin = ninf; umin = ninf; }
            Lmin[c] = lmin;
            meta[2 * c] = d; meta[2 * c + 1] = (int32_t)dbits;
            LU(c) = umin;
            Uc_mine = umin;
        }
        real_steps += __popcll(ballot64(avail));
        if (lane == 0) exp_lds[k] = leaf;
        n_nodes += A;
        k_done = k + 1;
        if (any64(bad)) { status = MP_ERR_REWARD_RANGE; break; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); // (one wave per workgroup: in-order memory operations; no wait for the stores' acknowledgements)
        {
            const int j = (lane - g) & 63;
            const double u = __shfl(Uc_mine, j & 63);
            if (j < A) {
                const int id = g + j;
                if (u > cbu) { cbu = u; cbid = id; }
            }
        }
    }
    }
    __syncthreads();

    if (status == MP_OK) {
        // ---- all backup_to_root calls at once: the bottom-up fixed point of L = max_c L_c, U = max_c U_c on the
        // scalars min_m (see opd.hip).  Upper bounds: leaves keep their own, the root's is the max over all leaves,
        // the other expanded nodes' are filled in by the export.
        double root_upper = ninf;
        for (int i = lane; i < n_nodes; i += 64) {
            const double u = LU(i);
            Umin[i] = u;
            if (u > root_upper) root_upper = u;
        }
        root_upper = wave_max(root_upper);
        __syncthreads();
        Pcg64 gen;
        gen.load(p.rng + (long)root * 6);
        int len = 0;
        double root_lower;
        if (!p.closing_chain && closing_compact_fits(p.K, A, p.cap, 64L * T * 8)) {
            // pointer jumping over the expansion tree + a prepared plan walk, on the scalars min_m L (opd_closing.hpp)
            len = closing_compact(lds, p.K, k_done, n_nodes, A, exp_lds, [&](int id) { return Lmin[id]; },
                                  [&](int id, double v) { Lmin[id] = v; }, gen,
                                  p.plans ? p.plans + (long)root * p.max_plan_len : nullptr, p.max_plan_len, root_lower);
        } else {
            for (int i = lane; i < n_nodes; i += 64) LU(i) = Lmin[i];
            __syncthreads();
            // chunks of 64 expansions from the last one down, one expansion per lane, each chunk repeated until no lane
            // computed a new maximum (opd.hip: the backups as a fixed point)
            for (int kb = (k_done - 1) & ~63; kb >= 0; kb -= 64) {
                const int k = kb + lane;
                const bool on = k < k_done;
                const int parent = on ? exp_lds[k] : 0;
                const int g = 1 + k * A;
                double last = __hiloint2double((int)0x7FF80000, 0); // NaN: the first repeat always writes
                for (;;) {
                    double m = ninf;
                    if (on)
                        for (int a = 0; a < A; ++a) {
                            const double l = LU(g + a);
                            m = l > m ? l : m;
                        }
                    const bool changed = on && !(m == last);
                    if (changed) { LU(parent) = m; last = m; }
                    __builtin_amdgcn_wave_barrier();
                    if (!any64(changed)) break;
                }
            }
            __syncthreads();
            for (int k = lane; k < k_done; k += 64) {
                const int n = exp_lds[k];
                Lmin[n] = LU(n);
            }
            // ---- get_plan with DeterministicNode.selection_rule over get_value_lower_bound = np.min
            // the bounds array becomes the node -> expansion-index map (NaN-boxed k in the slot of every expanded node): a
            // level of the descent is one round trip (the children's final min L from Lmin[], their slots from LDS), not a
            // search of the parent map (opd.hip)
            __syncthreads();
            root_lower = LU(0);
            __builtin_amdgcn_wave_barrier();
            for (int k = lane; k < k_done; k += 64) LU(exp_lds[k]) = __hiloint2double((int)0xFFF80000, k);
            __syncthreads();
            int kcur = k_done > 0 ? 0 : -1;
            while (kcur >= 0) {
                const int fc = 1 + kcur * A;
                const double l = lane < A ? Lmin[fc + lane] : ninf;
                const double slot = lane < A ? LU(fc + lane) : 0.0;
                const double m = A <= 16 ? row0_max(l) : wave_max(l);
                const unsigned long long ties = ballot64(lane < A && l == m);
                const int nt = __popcll(ties);
                int pick = (int)gen.below((uint32_t)nt);
                unsigned long long t = ties;
                while (pick-- > 0) t &= t - 1;
                const int a = __ffsll((long long)t) - 1;
                if (lane == 0 && p.plans && len < p.max_plan_len) p.plans[(long)root * p.max_plan_len + len] = a;
                ++len;
                const int shi = __builtin_amdgcn_readlane(__double2hiint(slot), a), slo = __builtin_amdgcn_readlane(__double2loint(slot), a);
                kcur = ((unsigned)shi == 0xFFF80000u) ? slo : -1; // expanded: its k; a leaf: the plan ends
            }
        }
        if (lane == 0) {
            gen.store(p.rng + (long)root * 6);
            if (p.plans)
                for (int i = len; i < p.max_plan_len; ++i) p.plans[(long)root * p.max_plan_len + i] = -1;
            if (p.plan_len) p.plan_len[root] = len;
            if (p.root_lower) p.root_lower[root] = root_lower;
            if (p.root_upper) p.root_upper[root] = root_upper;
        }
    } else if (lane == 0) {
        if (p.plans)
            for (int i = 0; i < p.max_plan_len; ++i) p.plans[(long)root * p.max_plan_len + i] = -1;
        if (p.plan_len) p.plan_len[root] = 0;
    }
    if (lane == 0) {
        if (p.status) p.status[root] = status;
        if (p.env_steps) p.env_steps[root] = (int64_t)real_steps; // one joint step per (real) child (deterministic.py:41)
        p.n_nodes_out[root] = n_nodes;
    }
    if (EXPG) {
        for (int k = k_done + lane; k < p.K; k += 64) p.expanded[(long)root * p.K + k] = -1;
    } else {
        for (int k = lane; k < p.K; k += 64) p.expanded[(long)root * p.K + k] = k < k_done ? exp_lds[k] : -1;
    }
#undef LU
}

// The high-occupancy form (opd.hip's opd_wide_kernel on the minima): bounds array in HBM/L2 with the key argmax, the
// selected leaf's slot NaN-boxed with its expansion index (node -> k map; the k -> node map is scattered by the closing
// pass; the plan descent needs no search), the closing lower-bound sweep through a sliding LDS window over Lmin[], and
// the arguments only the closing passes need loaded after the main loop (SGPR budget of 8 waves per SIMD).
template <bool NONNEG, bool SIB>
__global__ __launch_bounds__(64, 8) void ropd_wide_kernel(ROpdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // SIB: the sibling layout of opd.hip's opd_wide_kernel -- the |A| children of expansion k together, in group k + 1 of
    // row (k + 1) mod 64 (one write request per expansion instead of |A| scattered ones); a leaf is named by its code
    // (group << lgP) | child, which orders like the node ids; lane l caches the best of row l.
    const int T = SIB ? p.Tsib : p.T;
    const int lgP = p.lgP, P = 1 << lgP;
    double *leafU = p.leaf_global + (long)blockIdx.x * 64 * T;
#define LU(id) leafU[((id) & 63) * T + ((id) >> 6)]
    const int lane = threadIdx.x, root = blockIdx.x, A = p.A, M = p.M;
    const long base = (long)root * p.cap, SA = (long)p.S * A;
    double *Lv = p.Lv + base * M, *Rv = p.Rv + base * M, *Lmin = p.Lmin + base;
    int32_t *Sv = p.Sv + base * M, *meta = p.meta + base * 2;
    const uint32_t done_bit = p.done_on_next ? 2u : 1u;
    const double ninf = -INFINITY;

    if (lane == 0) {
        for (int m = 0; m < M; ++m) { Lv[m] = 0.0; Sv[m] = p.root_state[(long)root * M + m]; Rv[m] = 0.0; }
        Lmin[0] = 0.0; meta[0] = 0; meta[1] = 0;
        if (!SIB) LU(0) = 0.0;
    }
    if (SIB && lane < A) leafU[lane] = lane == 0 ? 0.0 : ninf; // group 0 of row 0: the root (its other slots are never leaves)
    __syncthreads();
    const bool small = (SIB ? T - 16 : T) <= 128; // at most two slots per lane in a re-scan
    const int cb0 = SIB ? ((((lane / A) << 6) << lgP) | (lane % A)) : 0;               // SIB: code of slot t = lane of row 0 ...
    const int cb1 = SIB ? (((((lane + 64) / A) << 6) << lgP) | ((lane + 64) % A)) : 0; // ... and of slot lane + 64
    int n_nodes = 1, status = MP_OK, k_done = 0;
    int real_steps = 0; // children of listed actions = planner.step calls of the joint environment
    double cbu = lane == 0 ? 0.0 : ninf;
    int cbid = lane == 0 ? 0 : 0x7fffffff;

    for (int k = 0; k < p.K; ++k) {
        // ---- robust.py:37: first maximal min_m U among the leaves
        double bu = cbu;
        int leaf = cbid;
        if (NONNEG) wave_argmax_keys_nonneg(bu, leaf); else wave_argmax_keys(bu, leaf);
        // (SIB: `leaf` is a code until here)
        const int el = SIB ? leaf >> lgP : 0, jl = SIB ? leaf & (P - 1) : 0;
        const int cls = SIB ? el & 63 : leaf & 63;
        if (SIB) leaf = el == 0 ? 0 : 1 + (el - 1) * A + jl;
        const int dleaf = meta[2 * leaf];
        if (lane == 0) { // dead to every selection; carries k
            if (SIB) leafU[cls * T + (el >> 6) * A + jl] = __hiloint2double((int)0xFFF80000, k);
            else LU(leaf) = __hiloint2double((int)0xFFF80000, k);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); // (one wave per workgroup: in-order memory operations; no wait for the stores' acknowledgements)
        {
            const double *row = leafU + cls * T;
            const int cnt = SIB ? (((k - cls) >> 6) + 1) * A : (n_nodes - cls + 63) >> 6; // SIB: the groups e <= k of this row
            const int cshift = cls << lgP;
            double ru = ninf;
            int rid = 0x7fffffff;
            if (small) { // at most two entries per lane (budget 5000: 79 per class): no loop, both reads unconditional
                const double u0 = row[lane < cnt ? lane : 0], u1 = row[lane + 64 < cnt ? lane + 64 : 0];
                if (lane < cnt && u0 > ru) { ru = u0; rid = SIB ? cb0 | cshift : cls + (lane << 6); }
                if (lane + 64 < cnt && u1 > ru) { ru = u1; rid = SIB ? cb1 | cshift : cls + ((lane + 64) << 6); }
            } else {
                for (int t = lane; t < cnt; t += 128) { // two reads in flight per trip
                    const double u0 = row[t];
                    const double u1 = t + 64 < cnt ? row[t + 64] : ninf;
                    if (u0 > ru) { ru = u0; rid = SIB ? ((((t / A) << 6) << lgP) | (t % A)) | cshift : cls + (t << 6); }
                    if (u1 > ru) { ru = u1; rid = SIB ? (((((t + 64) / A) << 6) << lgP) | ((t + 64) % A)) | cshift : cls + ((t + 64) << 6); }
                }
            }
            if (NONNEG) wave_argmax_keys_nonneg(ru, rid); else wave_argmax_keys(ru, rid);
            if (lane == cls) { cbu = ru; cbid = rid; }
        }
        // ---- DeterministicNode.expand (deterministic.py:28-43), update() with ndarray reward / done (:45-65)
        const int d = __builtin_amdgcn_readfirstlane(dleaf) + 1;
        typedef const double __attribute__((address_space(4))) *scalar_f64;
        const double g1d = ((scalar_f64)(unsigned long long)p.g1)[d], gdivd = ((scalar_f64)(unsigned long long)p.gdiv)[d],
                     tdivd = ((scalar_f64)(unsigned long long)p.tdiv)[d];
        const int g = n_nodes;
        bool bad = false, avail = false;
        double Uc_mine = 0.0;
        if (lane < A) {
            const int c = g + lane;
            double lmin = 0.0, umin = 0.0;
            uint32_t dbits = 0;
            const double *Lp = Lv + (long)leaf * M;
            const int32_t *Sp = Sv + (long)leaf * M;
            for (int m = 0; m < M; ++m) { // JointEnv.step: every model steps its own state (robust.py:13-16)
                const Rec rc = p.rec[(long)m * SA + (long)Sp[m] * A + lane];
                const double r = rc.reward;
                // robust.py:22-25: the joint env lists the UNION of the actions its models list in their own states
                avail |= (rc.flags & 4u) != 0;
                bad |= !(0.0 <= r) || !(r <= 1.0); // np.all(0 <= reward), np.all(reward <= 1)
                const bool dn = (rc.flags & done_bit) != 0;
                double Lc = Lp[m] + g1d * r;
                double Uc = Lc + gdivd;
                if (dn) {
                    const double nv = Lc + tdivd;
                    Lc = nv; Uc = nv;
                }
                Lv[(long)c * M + m] = Lc;
                Sv[(long)c * M + m] = rc.next;
                Rv[(long)c * M + m] = (m >= 32 && dn) ? reward_with_done(r) : r; // (models beyond the 32 done bits of the node: the flag rides in bit 62 of the reward)
                if (m < 32) dbits |= (dn ? 1u : 0u) << m;
                if (m == 0 || Lc < lmin) lmin = Lc; // np.min
                if (m == 0 || Uc < umin) umin = Uc;
            }
            // deterministic.py:32-35: an action no model lists gets no child; its slot stays in the id space as a PHANTOM
            // with min L = min U = -inf (never the best leaf, never a maximum of the backup, never a tie of the plan;
            // dropped by the export) -- see opd.hip
            bad = bad && avail;
            if (!avail) { lmin = ninf; umin = ninf; }
            Lmin[c] = lmin;
            meta[2 * c] = d; meta[2 * c + 1] = (int32_t)dbits;
            if (SIB) leafU[((k + 1) & 63) * T + ((k + 1) >> 6) * A + lane] = umin; // the whole group in one request
            else LU(c) = umin;
            Uc_mine = umin;
        }
        real_steps += __popcll(ballot64(avail));
        n_nodes += A;
        k_done = k + 1;
        if (any64(bad)) { status = MP_ERR_REWARD_RANGE; break; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); // (one wave per workgroup: in-order memory operations; no wait for the stores' acknowledgements)
        if (SIB) { // the children, all of ONE row, against that row's best (lane re)
            const int e = k + 1, re = e & 63;
            const double cbu_re = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cbu), re), __builtin_amdgcn_readlane(__double2loint(cbu), re));
            const bool mine = lane < A;
            if (any64(mine && Uc_mine > cbu_re)) { // (wave-uniform)
                const double um = mine ? Uc_mine : ninf;
                const double m = A <= 16 ? row0_max(um) : wave_max(um);
                const int jm = __ffsll((long long)ballot64(mine && Uc_mine == m)) - 1; // lowest id among equal bounds
                if (lane == re) { cbu = m; cbid = (e << lgP) + jm; }
            }
        } else {
            const int j = (lane - g) & 63;
            const double u = __shfl(Uc_mine, j & 63);
            if (j < A) {
                const int id = g + j;
                if (u > cbu) { cbu = u; cbid = id; }
            }
        }
    }
    __syncthreads();

    const ROpdArgs __attribute__((address_space(4))) *q; // closing-only arguments: read here, not held through the loop
    {
        unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        q = (const ROpdArgs __attribute__((address_space(4))) *)ka;
    }
    double *Umin = q->Umin + base;
    int32_t *exp_map = q->expanded + (long)root * (q->K > 0 ? q->K : 1);
    int32_t *const plans = q->plans, *const plan_len = q->plan_len;
    const int max_plan_len = q->max_plan_len;

    double root_upper = ninf;
    if (SIB) {
        for (int r = 0; r < 64 && r <= k_done; ++r) {
            const int cnt = (((k_done - r) >> 6) + 1) * A;
            for (int t = lane; t < cnt; t += 64) { // slot t = group t / |A| of row r, child t mod |A|
                const int e = ((t / A) << 6) | r, jj = t % A;
                if (e > 0 || jj == 0) {
                    const int id = e == 0 ? 0 : 1 + (e - 1) * A + jj;
                    const double u = leafU[r * T + t];
                    Umin[id] = u != u ? ninf : u;
                    if (u > root_upper) root_upper = u;
                    if (u != u) exp_map[__double2loint(u)] = id;
                }
            }
        }
    } else {
        for (int i = lane; i < n_nodes; i += 64) {
            const double u = LU(i);
            Umin[i] = u != u ? ninf : u;
            if (u > root_upper) root_upper = u;
            if (u != u) exp_map[__double2loint(u)] = i;
        }
    }
    root_upper = wave_max(root_upper);
    __syncthreads();

    if (status == MP_OK) {
        double *win = lds;
        const int C = q->chunk;
        for (int k0 = k_done - 1; k0 >= 0;) {
            const int kb = k0 & ~(C - 1);
            const int lo = 1 + kb * A, n_win = (k0 - kb + 1) * A;
            __syncthreads();
            for (int i = lane; i < n_win; i += 64) win[i] = Lmin[lo + i];
            const int ek = lane <= k0 - kb ? exp_map[kb + lane] : 0;
            __syncthreads();
            for (int k = k0; k >= kb; --k) {
                const int g = (k - kb) * A;
                const double mine = lane < A ? win[g + lane] : ninf;
                const double m = A <= 16 ? row0_max(mine) : wave_max(mine);
                const int parent_k = __builtin_amdgcn_readlane(ek, k - kb);
                if (lane == 0) {
                    Lmin[parent_k] = m;
                    if (parent_k >= lo) win[parent_k - lo] = m;
                }
                __builtin_amdgcn_wave_barrier();
            }
            k0 = kb - 1;
        }
        __syncthreads();
        Pcg64 gen;
        gen.load(q->rng + (long)root * 6);
        int len = 0;
        int kcur = k_done > 0 ? 0 : -1;
        while (kcur >= 0) {
            const int fc = 1 + kcur * A;
            const double l = lane < A ? Lmin[fc + lane] : ninf;
            const double slot = lane >= A ? 0.0 : SIB ? leafU[((kcur + 1) & 63) * T + ((kcur + 1) >> 6) * A + lane] : LU(fc + lane);
            const double m = A <= 16 ? row0_max(l) : wave_max(l);
            const unsigned long long ties = ballot64(lane < A && l == m);
            const int nt = __popcll(ties);
            int pick = (int)gen.below((uint32_t)nt);
            unsigned long long t = ties;
            while (pick-- > 0) t &= t - 1;
            const int a = __ffsll((long long)t) - 1;
            if (lane == 0 && plans && len < max_plan_len) plans[(long)root * max_plan_len + len] = a;
            ++len;
            const int shi = __builtin_amdgcn_readlane(__double2hiint(slot), a), slo = __builtin_amdgcn_readlane(__double2loint(slot), a);
            kcur = ((unsigned)shi == 0xFFF80000u) ? slo : -1;
        }
        if (lane == 0) {
            gen.store(q->rng + (long)root * 6);
            if (plans)
                for (int i = len; i < max_plan_len; ++i) plans[(long)root * max_plan_len + i] = -1;
            if (plan_len) plan_len[root] = len;
            if (q->root_lower) q->root_lower[root] = Lmin[0];
            if (q->root_upper) q->root_upper[root] = root_upper;
        }
    } else if (lane == 0) {
        if (plans)
            for (int i = 0; i < max_plan_len; ++i) plans[(long)root * max_plan_len + i] = -1;
        if (plan_len) plan_len[root] = 0;
    }
    if (lane == 0) {
        if (q->status) q->status[root] = status;
        if (q->env_steps) q->env_steps[root] = (int64_t)real_steps;
        q->n_nodes_out[root] = n_nodes;
    }
    for (int k = k_done + lane; k < q->K; k += 64) exp_map[k] = -1;
#undef LU
}

// ---- any number of actions (|A| > 64): the plain form (opd.hip's opd_any_kernel on the minima over the models) -- one root
// per wavefront, everything in global memory, children 64 at a time, the leaf argmax a scan of min_m U.
__global__ __launch_bounds__(64) void ropd_any_kernel(ROpdArgs p)
{
    const int lane = threadIdx.x, root = blockIdx.x, A = p.A, M = p.M;
    const long base = (long)root * p.cap, SA = (long)p.S * A;
    double *Lv = p.Lv + base * M, *Rv = p.Rv + base * M, *Lmin = p.Lmin + base, *Umin = p.Umin + base;
    int32_t *Sv = p.Sv + base * M, *meta = p.meta + base * 2;
    int32_t *EXP = p.expanded + (long)root * (p.K > 0 ? p.K : 1);
    const uint32_t done_bit = p.done_on_next ? 2u : 1u;
    const double ninf = -INFINITY;
    if (lane == 0) {
        for (int m = 0; m < M; ++m) { Lv[m] = 0.0; Sv[m] = p.root_state[(long)root * M + m]; Rv[m] = 0.0; }
        Lmin[0] = 0.0; meta[0] = 0; meta[1] = 0;
        Umin[0] = 0.0;
    }
    __syncthreads();
    int n_nodes = 1, status = MP_OK, k_done = 0, real_steps = 0;
    for (int k = 0; k < p.K; ++k) {
        double bu = ninf; // robust.py:37: first maximal min_m U among the leaves (an expanded node's slot holds -inf)
        int leaf = 0x7fffffff;
        for (int i = lane; i < n_nodes; i += 64) {
            const double u = Umin[i];
            if (u > bu) { bu = u; leaf = i; }
        }
        wave_argmax(bu, leaf);
        if (leaf == 0x7fffffff) { status = MP_ERR_ARG; break; }
        const int d = meta[2 * leaf] + 1;
        const double g1d = p.g1[d], gdivd = p.gdiv[d], tdivd = p.tdiv[d];
        const int g = n_nodes;
        bool bad_any = false;
        const double *Lp = Lv + (long)leaf * M;
        const int32_t *Sp = Sv + (long)leaf * M;
        for (int a0 = 0; a0 < A; a0 += 64) {
            const int a = a0 + lane;
            bool bad = false, avail = false;
            if (a < A) {
                const int c = g + a;
                double lmin = 0.0, umin = 0.0;
                uint32_t dbits = 0;
                for (int m = 0; m < M; ++m) { // JointEnv.step: every model steps its own state (robust.py:13-16)
                    const Rec rc = p.rec[(long)m * SA + (long)Sp[m] * A + a];
                    const double r = rc.reward;
                    avail |= (rc.flags & 4u) != 0;     // robust.py:22-25: the union of what the models list
                    bad |= !(0.0 <= r) || !(r <= 1.0); // np.all(0 <= reward), np.all(reward <= 1)
                    const bool dn = (rc.flags & done_bit) != 0;
                    double Lc = Lp[m] + g1d * r;
                    double Uc = Lc + gdivd;
                    if (dn) {
                        const double nv = Lc + tdivd;
                        Lc = nv; Uc = nv;
                    }
                    Lv[(long)c * M + m] = Lc;
                    Sv[(long)c * M + m] = rc.next;
                    Rv[(long)c * M + m] = (m >= 32 && dn) ? reward_with_done(r) : r; // (models beyond the 32 done bits of the node: the flag rides in bit 62 of the reward)
                    if (m < 32) dbits |= (dn ? 1u : 0u) << m;
                    if (m == 0 || Lc < lmin) lmin = Lc; // np.min
                    if (m == 0 || Uc < umin) umin = Uc;
                }
                bad = bad && avail;
                if (!avail) { lmin = ninf; umin = ninf; } // a phantom slot (see ropd_kernel)
                Lmin[c] = lmin;
                meta[2 * c] = d; meta[2 * c + 1] = (int32_t)dbits;
                Umin[c] = umin;
            }
            real_steps += __popcll(ballot64(avail));
            bad_any |= any64(bad);
        }
        if (lane == 0) { Umin[leaf] = ninf; EXP[k] = leaf; }
        n_nodes += A;
        k_done = k + 1;
        if (bad_any) { status = MP_ERR_REWARD_RANGE; break; }
        __syncthreads();
    }
    __syncthreads();
    if (status == MP_OK) {
        double root_upper = ninf;
        for (int i = lane; i < n_nodes; i += 64) {
            const double u = Umin[i];
            if (u > root_upper) root_upper = u;
        }
        root_upper = wave_max(root_upper);
        for (int k = k_done - 1; k >= 0; --k) { // the backups on the scalars min_m L, children before parents
            const int fc = 1 + k * A;
            double m = ninf;
            for (int a = lane; a < A; a += 64) {
                const double l = Lmin[fc + a];
                m = l > m ? l : m;
            }
            m = wave_max(m);
            if (lane == 0) Lmin[EXP[k]] = m;
            __syncthreads();
        }
        Pcg64 gen;
        gen.load(p.rng + (long)root * 6);
        int len = 0, node = 0;
        for (;;) { // get_plan with DeterministicNode.selection_rule over get_value_lower_bound = np.min
            int kcur = -1;
            for (int k0 = 0; k0 < k_done && kcur < 0; k0 += 64) {
                const unsigned long long hit = ballot64(k0 + lane < k_done && EXP[k0 + lane] == node);
                if (hit) kcur = k0 + __ffsll((long long)hit) - 1;
            }
            if (kcur < 0) break;
            const int fc = 1 + kcur * A;
            double m = ninf;
            for (int a = lane; a < A; a += 64) {
                const double l = Lmin[fc + a];
                m = l > m ? l : m;
            }
            m = wave_max(m);
            int nt = 0;
            for (int a0 = 0; a0 < A; a0 += 64) nt += __popcll(ballot64(a0 + lane < A && Lmin[fc + a0 + lane] == m));
            int pick = (int)gen.below((uint32_t)nt), act = 0;
            for (int a0 = 0; a0 < A; a0 += 64) {
                unsigned long long t = ballot64(a0 + lane < A && Lmin[fc + a0 + lane] == m);
                const int c = __popcll(t);
                if (pick < c) {
                    while (pick-- > 0) t &= t - 1;
                    act = a0 + __ffsll((long long)t) - 1;
                    break;
                }
                pick -= c;
            }
            if (lane == 0 && p.plans && len < p.max_plan_len) p.plans[(long)root * p.max_plan_len + len] = act;
            ++len;
            node = fc + act;
        }
        if (lane == 0) {
            gen.store(p.rng + (long)root * 6);
            if (p.plans)
                for (int i = len; i < p.max_plan_len; ++i) p.plans[(long)root * p.max_plan_len + i] = -1;
            if (p.plan_len) p.plan_len[root] = len;
            if (p.root_lower) p.root_lower[root] = Lmin[0];
            if (p.root_upper) p.root_upper[root] = root_upper;
        }
    } else if (lane == 0) {
        if (p.plans)
            for (int i = 0; i < p.max_plan_len; ++i) p.plans[(long)root * p.max_plan_len + i] = -1;
        if (p.plan_len) p.plan_len[root] = 0;
    }
    if (lane == 0) {
        if (p.status) p.status[root] = status;
        if (p.env_steps) p.env_steps[root] = (int64_t)real_steps;
        p.n_nodes_out[root] = n_nodes;
    }
    for (int k = k_done + lane; k < p.K; k += 64) EXP[k] = -1;
}

} // namespace mp

using namespace mp;

extern "C" {

int mp_ropd_plan(mp_ctx *ctx, mp_model *model, int32_t n_roots, const int32_t *root_state, int32_t budget, double gamma,
                 double terminal_reward, uint64_t *rng_state, int32_t max_plan_len, int32_t *plans, int32_t *plan_len,
                 double *root_lower, double *root_upper, int64_t *env_steps, int32_t *status, int32_t mem)
{
    if (!ctx || !model || !root_state || !rng_state) return fail(MP_ERR_ARG, "mp_ropd_plan: NULL argument");
    if (!mem_valid(mem)) return fail(MP_ERR_ARG, "mp_ropd_plan: unknown mem flags %d", mem);
    const int rmem = mem_rng(mem); // MP_MEM_RNG_DEVICE: the generator records are device-resident (mp_rng) also with host arrays
    mem = mem_arrays(mem);
    if (model->mode != MP_MODE_DETERMINISTIC || !model->rec_all)
        return fail(MP_ERR_MODE, "mp_ropd_plan: needs a joint model (mp_model_load_joint)");
    const int A = model->A, M = model->M;
    const bool any_a = A > 64; // more actions than lanes: the plain kernel (ropd_any_kernel)
    if (n_roots < 1 || budget < 0 || max_plan_len < 0) return fail(MP_ERR_ARG, "mp_ropd_plan: bad sizes");
    const int K = budget / A; // deterministic.py:118
    if (K > 0 && !(gamma != 1.0))
        return fail(MP_ERR_ARG, "mp_ropd_plan: gamma = 1 (the reference divides by 1 - gamma, deterministic.py:53)");
    const long cap = 1 + (long)K * A;
    const int T = (int)((cap + 63) / 64) | 1;
    const size_t lds_bounds = (size_t)64 * T * sizeof(double); // bounds only, parent map in HBM (EXPG)
    const size_t lds_full = lds_bounds + (size_t)(K > 0 ? K : 1) * sizeof(int32_t);
    int chunk = 64; // high-occupancy variant: LDS only holds the window of the closing pass
    while (chunk > 1 && (size_t)chunk * A * sizeof(double) > 4096) chunk >>= 1;
    const size_t lds_win = (size_t)chunk * A * sizeof(double);
    const char *force = getenv("MP_OPD_MODEL"); // "lds" / "ldsx" / "global": test hook (shared with mp_opd_plan)
    const long cus = ctx->prop.multiProcessorCount;
    const long lds_roots = cus * (long)((kLdsBytes - 1024) / lds_full);
    const long expg_roots = cus * (long)((kLdsBytes - 1024) / lds_bounds);
    bool glb = lds_bounds > kLdsBytes - 1024 || n_roots > expg_roots;
    if (force && force[0] == 'g') glb = true;
    if (force && force[0] == 'l' && lds_bounds <= kLdsBytes - 1024) glb = false;
    bool expg = !glb && (lds_full > kLdsBytes - 1024 || n_roots > lds_roots);
    if (force && !glb && force[1] == 'd' && force[2] == 's' && force[3] == 'x') expg = true;
    const size_t lds = glb ? lds_win : (expg ? lds_bounds : lds_full);
    MP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    const int D = K + 2;
    std::vector<double> tab((size_t)3 * D);
    for (int d = 0; d < D; ++d) { // host libm, bit-equal to Python's float ** (as in mp_opd_plan)
        tab[d] = d >= 1 ? pow(gamma, (double)(d - 1)) : 0.0;
        tab[D + d] = pow(gamma, (double)d) / (1 - gamma);
        tab[2 * D + d] = terminal_reward * pow(gamma, (double)d) / (1 - gamma);
    }
    double *d_tab = nullptr;
    MP_TRY(upload_tables(ctx, 2, tab, &d_tab));

    ROpdArgs a;
    // high-occupancy variant: the sibling layout of its bounds array (default; MP_OPD_WIDE=cls: the residue-class layout), see opd.hip
    int lgP = 0;
    while ((1 << lgP) < A) ++lgP;
    const int Tsib = ((K + 1 + 63) / 64) * A + 16;
    const char *wide_env = getenv("MP_OPD_WIDE");
    const bool sib = !(wide_env && wide_env[0] == 'c');
    a.n_roots = n_roots; a.M = M; a.S = model->S; a.A = A; a.K = K; a.cap = (int)cap; a.T = T; a.chunk = chunk; a.Tsib = Tsib; a.lgP = lgP;
    { const char *cl = getenv("MP_OPD_CLOSING"); a.closing_chain = cl && cl[0] == 'c'; }
    a.done_on_next = model->done_on_next; a.max_plan_len = max_plan_len;
    a.rec = model->rec_all;
    a.g1 = d_tab; a.gdiv = d_tab + D; a.tdiv = d_tab + 2 * D;
    const size_t nn = (size_t)n_roots * cap;
    MP_TRY(ws_get(ctx, WS_TREE0, nn * M, &a.Lv));
    MP_TRY(ws_get(ctx, WS_TREE1, nn * M, &a.Sv));
    MP_TRY(ws_get(ctx, WS_TREE2, nn, &a.Lmin));
    MP_TRY(ws_get(ctx, WS_TREE3, nn, &a.Umin));
    MP_TRY(ws_get(ctx, WS_TREE4, nn * M, &a.Rv));
    MP_TRY(ws_get(ctx, WS_TREE5, nn * 2, &a.meta));
    a.leaf_global = nullptr;
    if (glb && !any_a) MP_TRY(ws_get(ctx, WS_TREE6, (size_t)n_roots * 64 * (sib ? Tsib : T), &a.leaf_global));
    MP_TRY(ws_get(ctx, WS_TREE7, (size_t)n_roots * (K > 0 ? K : 1) + n_roots, &a.expanded));
    a.n_nodes_out = a.expanded + (size_t)n_roots * (K > 0 ? K : 1);
    ctx->tree.kind = 3; ctx->tree.n_roots = n_roots; ctx->tree.A = A; ctx->tree.cap = (int)cap; ctx->tree.K = K;
    ctx->tree.M = M; ctx->tree.gamma = gamma;

    int32_t *d_rs = nullptr;
    MP_TRY(stage_in(ctx, WS_IO0, root_state, (size_t)n_roots * M, mem, &d_rs));
    a.root_state = d_rs;
    MP_TRY(stage_in(ctx, WS_IO2, (const uint64_t *)rng_state, (size_t)n_roots * 6, rmem, &a.rng));
    MP_TRY(stage_out_alloc(ctx, WS_IO3, plans, (size_t)n_roots * max_plan_len, mem, &a.plans));
    MP_TRY(stage_out_alloc(ctx, WS_IO4, plan_len, (size_t)n_roots, mem, &a.plan_len));
    MP_TRY(stage_out_alloc(ctx, WS_IO5, root_lower, (size_t)n_roots, mem, &a.root_lower));
    MP_TRY(stage_out_alloc(ctx, WS_IO6, root_upper, (size_t)n_roots, mem, &a.root_upper));
    MP_TRY(stage_out_alloc(ctx, WS_IO7, status, (size_t)n_roots, mem, &a.status));
    MP_TRY(stage_out_alloc(ctx, WS_IO8, env_steps, (size_t)n_roots, mem, &a.env_steps));

    // main loop: the batched-load form for <= 2 / <= 4 models when every finite bound is >= +0.0, else the generic one
    // (MP_OPD_LOOP=0: the generic one always -- test hook, shared with mp_opd_plan)
    const char *mode_env = getenv("MP_OPD_LOOP");
    int mb = gamma >= 0 && gamma < 1 && terminal_reward >= 0 && !(mode_env && mode_env[0] == '0') ? (M <= 2 ? 2 : M <= 4 ? 4 : 0) : 0;
    typedef void (*kernel_t)(ROpdArgs);
    const kernel_t kfn = expg ? (mb == 2 ? ropd_kernel<true, 2> : mb == 4 ? ropd_kernel<true, 4> : ropd_kernel<true, 0>)
                              : (mb == 2 ? ropd_kernel<false, 2> : mb == 4 ? ropd_kernel<false, 4> : ropd_kernel<false, 0>);
    if (!any_a && lds > 64 * 1024)
        MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MP_TRY(kernels_begin(ctx));
    const bool nonneg = gamma >= 0 && gamma < 1 && terminal_reward >= 0 && !(mode_env && mode_env[0] == '0');
    if (any_a) hipLaunchKernelGGL(ropd_any_kernel, dim3((unsigned)n_roots), dim3(64), 0, st, a);
    else if (glb) {
        const kernel_t kw = sib ? (nonneg ? ropd_wide_kernel<true, true> : ropd_wide_kernel<false, true>)
                                : (nonneg ? ropd_wide_kernel<true, false> : ropd_wide_kernel<false, false>);
        hipLaunchKernelGGL(kw, dim3((unsigned)n_roots), dim3(64), lds, st, a);
    }
    else hipLaunchKernelGGL(kfn, dim3((unsigned)n_roots), dim3(64), lds, st, a);
    MP_TRY(kernels_end(ctx, 1));
    MP_HIP(hipGetLastError());

    MP_TRY(stage_out_copy(ctx, rng_state, a.rng, (size_t)n_roots * 6, rmem));
    MP_TRY(stage_out_copy(ctx, plans, a.plans, (size_t)n_roots * max_plan_len, mem));
    MP_TRY(stage_out_copy(ctx, plan_len, a.plan_len, (size_t)n_roots, mem));
    MP_TRY(stage_out_copy(ctx, root_lower, a.root_lower, (size_t)n_roots, mem));
    MP_TRY(stage_out_copy(ctx, root_upper, a.root_upper, (size_t)n_roots, mem));
    MP_TRY(stage_out_copy(ctx, status, a.status, (size_t)n_roots, mem));
    MP_TRY(stage_out_copy(ctx, env_steps, a.env_steps, (size_t)n_roots, mem));
    if (mem == MP_MEM_HOST) MP_HIP(hipStreamSynchronize(st));
    return MP_OK;
}

int mp_ropd_tree_export(mp_ctx *ctx, int32_t root, int32_t cap, int32_t *n_nodes, int32_t *parent, int32_t *action,
                        int32_t *state, int32_t *depth, double *reward, double *lower, double *upper, uint8_t *done,
                        int64_t *count, int32_t *first_child, int32_t *n_children)
{
    if (!ctx) return fail(MP_ERR_ARG, "ctx is NULL");
    if (ctx->tree.kind != 3) return fail(MP_ERR_ARG, "mp_ropd_tree_export: no robust OPD tree on this ctx");
    if (root < 0 || root >= ctx->tree.n_roots) return fail(MP_ERR_ARG, "mp_ropd_tree_export: root %d out of range", root);
    const int tcap = ctx->tree.cap, A = ctx->tree.A, K = ctx->tree.K, NR = ctx->tree.n_roots, M = ctx->tree.M;
    const double gamma = ctx->tree.gamma;
    MP_HIP(hipSetDevice(ctx->device));
    MP_HIP(hipStreamSynchronize(ctx->stream));
    const int32_t *d_exp = (const int32_t *)ctx->ws[WS_TREE7].p;
    int32_t n = 0;
    MP_HIP(hipMemcpy(&n, d_exp + (size_t)NR * (K > 0 ? K : 1) + root, sizeof(int32_t), hipMemcpyDeviceToHost));
    const long base = (long)root * tcap;
    auto pull = [&](void *dst, int slot, size_t elt) -> int {
        MP_HIP(hipMemcpy(dst, (const char *)ctx->ws[slot].p + base * elt, (size_t)n * elt, hipMemcpyDeviceToHost));
        return MP_OK;
    };
    std::vector<double> lv((size_t)n * M), rv((size_t)n * M), lmin((size_t)n), umin((size_t)n);
    std::vector<int32_t> sv((size_t)n * M), meta((size_t)n * 2), fc((size_t)n, -1), exp((size_t)(K > 0 ? K : 1)), par((size_t)n);
    MP_TRY(pull(lv.data(), WS_TREE0, sizeof(double) * M));
    MP_TRY(pull(sv.data(), WS_TREE1, sizeof(int32_t) * M));
    MP_TRY(pull(lmin.data(), WS_TREE2, sizeof(double)));
    MP_TRY(pull(umin.data(), WS_TREE3, sizeof(double)));
    MP_TRY(pull(rv.data(), WS_TREE4, sizeof(double) * M));
    MP_TRY(pull(meta.data(), WS_TREE5, sizeof(int32_t) * 2));
    MP_HIP(hipMemcpy(exp.data(), d_exp + (size_t)root * (K > 0 ? K : 1), (size_t)(K > 0 ? K : 1) * sizeof(int32_t),
                     hipMemcpyDeviceToHost));
    for (int k = 0; k < K && 1 + (k + 1) * A <= n; ++k)
        if (exp[k] >= 0 && exp[k] < n) fc[exp[k]] = 1 + k * A;
    par[0] = -1;
    for (int i = 1; i < n; ++i) par[i] = exp[(i - 1) / A];
    for (int i = n - 1; i >= 0; --i) // expanded nodes: U = max over children of their scalar (bottom-up)
        if (fc[i] >= 0) {
            double m = umin[fc[i]];
            for (int a = 1; a < A; ++a)
                if (umin[fc[i] + a] > m) m = umin[fc[i] + a];
            umin[i] = m;
        }
    // slots of actions no model lists (deterministic.py:32-35 over JointEnv.get_available_actions) are phantoms with
    // min L = -inf: not nodes of the tree
    std::vector<int32_t> id((size_t)n, -1);
    int kept = 0;
    for (int i = 0; i < n; ++i)
        if (!(lmin[i] == -INFINITY)) id[i] = kept++;
    if (kept > cap) return fail(MP_ERR_ARG, "mp_ropd_tree_export: capacity %d < %d nodes", cap, kept);
    std::vector<int64_t> sz((size_t)n, 0);
    for (int i = n - 1; i >= 0; --i) {
        if (id[i] < 0) continue;
        sz[i] += 1;
        if (i > 0) sz[par[i]] += sz[i];
    }
    for (int i = 0; i < n; ++i) {
        if (id[i] < 0) continue;
        const int o = id[i];
        const int d = meta[2 * i];
        if (parent) parent[o] = i == 0 ? -1 : id[par[i]];
        if (action) action[o] = i == 0 ? -1 : (i - 1) % A;
        if (depth) depth[o] = d;
        if (count) count[o] = i == 0 ? sz[0] : 1 + sz[i];
        int first = -1, nc = 0;
        if (fc[i] >= 0)
            for (int a = 0; a < A; ++a) {
                const int c = id[fc[i] + a];
                if (c < 0) continue;
                if (first < 0) first = c;
                ++nc;
            }
        if (first_child) first_child[o] = first;
        if (n_children) n_children[o] = nc;
        for (int m = 0; m < M; ++m) {
            const size_t j = (size_t)i * M + m, jo = (size_t)o * M + m;
            const bool dn = m < 32 ? (((uint32_t)meta[2 * i + 1] >> m) & 1u) != 0 : reward_done(rv[j]);
            if (state) state[jo] = sv[j];
            if (reward) reward[jo] = m < 32 ? rv[j] : reward_plain(rv[j]);
            if (done) done[jo] = (uint8_t)dn;
            // a leaf keeps its vectors (U recomputed as update() computed it, deterministic.py:51-59: same host
            // operations as the planning tables); an expanded node holds the backed-up scalars
            if (lower) lower[jo] = fc[i] >= 0 ? lmin[i] : lv[j];
            if (upper) upper[jo] = fc[i] >= 0 ? umin[i] : (i == 0 ? 0.0 : (dn ? lv[j] : lv[j] + pow(gamma, (double)d) / (1 - gamma)));
        }
    }
    if (n_nodes) *n_nodes = kept;
    return MP_OK;
}

} // extern "C"
