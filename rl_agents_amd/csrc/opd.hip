// opd.hip -- optimistic planning for deterministic systems (tree_search/deterministic.py).
//
// Mapping: ONE ROOT PER WAVEFRONT (one 64-lane workgroup per root).  OPD is sequential per root
// (each expansion depends on the previous backup) but every expansion starts with an argmax over
// all current leaves -- `max(self.leaves, key=U)`, deterministic.py:110, O(#leaves) in the
// reference and 44 % of its run time -- which is wave-parallel:
//   * the upper bound of every node is kept in LDS, indexed by node id (creation order), with
//     expanded nodes overwritten by -inf; 64 lanes scan it with ds_read_b64 and reduce with
//     cross-lane shuffles on (U, id), maximal U first, lowest id among equals -- exactly the
//     "first maximal element in list order" the reference picks, because its leaves list is in
//     creation order (removal keeps order, children are appended, deterministic.py:29,42).
//   * the |A| children are created by |A| lanes: one 16-byte model gather each, bounds from
//     host-computed gamma-power tables (libm pow = Python **), node records to HBM.
//   * backup_to_root (deterministic.py:74-79) is NOT replayed per expansion.  No decision made
//     during planning reads an internal node's bounds (leaf selection reads leaf U, a child's L
//     starts from its parent's creation-time L), and after every reference backup each ancestor
//     equals the max over its children; so the final bounds are the unique bottom-up fixed point
//     L[n] = max_c L[c], U[n] = max_c U[c].  L is computed once at the end, in reverse expansion
//     order (children always have larger ids than their parent), inside LDS; the root's U is one
//     reduction over the leaves and the other internal U's are filled in at export: O(K) instead of the
//     reference's O(K * depth) -- on highway-shaped tables, whose reward-1 lane makes the tree a
//     chain, that removes 62 dependent HBM round trips per expansion.  max is exact, so the
//     bounds are bit-identical to the incremental walk.
//   * `count` (deterministic.py:62-63) is not needed by any decision; it is reconstructed from
//     subtree sizes at export time.
// HBM per root: {L f64, state i32, depth i32 | done << 30} 16 B + U f64 + reward f64; first_child is derived from the
// parent map by the export.  Every vector memory instruction costs the texture-address unit its slot whatever it
// moves (TA busy 72-84 % at 8192 roots), so an expansion issues as few as possible: one record store per child
// instead of four, gamma tables through the scalar cache (the depth is wave-uniform)
// = 37 B/node; LDS per root: 8 B/node (+ 4 B per expansion for the parent map).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.hpp"
#include "pcg64.hpp"
#include "wave.hpp"
#include "opd_closing.hpp"

// The selected leaf's 16-byte node record, read by a SCALAR load past the scalar cache (glc).  The record was written by this
// wave's vector stores of an EARLIER expansion; `s_waitcnt vmcnt(N)` with N = the vector-memory operations THIS expansion has
// issued so far (the tag store into leafU + the row loads) waits for exactly those older stores without stalling on the row
// loads just issued.  N is hand-counted against the code hipcc emits: rl_agents_amd/build.py disassembles opd.o after every
// build and REFUSES the library unless the N vector-memory instructions preceding each of these waits are the expected
// {N - 1 x global_load_dwordx2, 1 x global_store_dwordx2} (tools/check_isa.py).  -DMP_OPD_SAFE_WAITCNT waits for
// everything instead (vmcnt(0): correct under any code generation, about one L2 round trip slower per expansion).
#ifdef MP_OPD_SAFE_WAITCNT
#define OPD_LEAF_SLOAD(N, lr, lp) \
    asm volatile("s_waitcnt vmcnt(0)\n\ts_load_dwordx4 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(lr) : "s"(lp) : "memory")
#else
#define OPD_LEAF_SLOAD(N, lr, lp) \
    asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_load_dwordx4 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(lr) : "s"(lp) : "memory")
#endif

namespace mp {

struct OpdArgs {
    int n_roots, S, A, K, cap, done_on_next, max_plan_len;
    int T; // row length of a residue class: odd, >= ceil(cap / 64)
    int Tsib, lgP; // opd_wide_kernel<SIB>: row length of the sibling layout; a leaf's code is (group << lgP) | child, 2^lgP >= |A|
    int chunk; // opd_wide_kernel: expansions per LDS window of the closing lower-bound pass (power of two <= 64)
    int closing_chain; // opd_kernel: 1 = the node-array closing passes even where opd_closing.hpp fits (MP_OPD_CLOSING=chain: test hook)
    const Rec *rec;
    const int32_t *root_state;
    const double *g1;   // g1[d]   = gamma ** (d - 1), d >= 1
    const double *gdiv; // gdiv[d] = gamma ** d / (1 - gamma)
    const double *tdiv; // tdiv[d] = terminal_reward * gamma ** d / (1 - gamma)
    uint64_t *rng;
    // per-root node arrays, root-major [n_roots][cap]
    double *L; // OpdNode records {L, state, depth}, 16 B per node
    double *U, *reward;
    double *leaf_global; // [n_roots][64 * T]: the upper-bound array of the high-occupancy variant (else nullptr)
    int32_t *expanded; // [n_roots][K] node expanded at step k (= parent of nodes 1 + kA .. 1 + kA + A - 1)
    int32_t *n_nodes_out;
    int32_t *plans, *plan_len, *status;
    double *root_lower, *root_upper;
    int64_t *env_steps;
};

struct alignas(16) OpdNode {
    double L;      // lower bound (creation-time value for leaves; final value after the bottom-up pass)
    int32_t state; // cloned-environment state
    int32_t depth;
};
static_assert(sizeof(OpdNode) == 16, "OpdNode must be one dwordx4");

// LDS layout of the upper-bound array: node id -> (id & 63) * T + (id >> 6).  Lane l owns the ids
// congruent to l mod 64 ("class" l), stored contiguously, and caches the best leaf of its class in
// registers.  An expansion then costs: one 64-lane argmax over the cached class maxima, one
// cooperative re-scan of the winner's class only (cap/64 entries, contiguous, conflict-free), and
// one compare per lane for the new children -- instead of a scan of all cap entries.
// opd_kernel<EXPG = false>: the upper-bound array AND the parent map live in LDS (44 KB per root at budget 5000 -> 3 roots
//              per CU: lowest latency per root, the choice while all roots of the batch are resident anyway).
// opd_kernel<EXPG = true>: the parent map goes to HBM (it is written once per expansion, off the chain, and read back in
//              coalesced chunks by the closing passes), leaving 64*T*8 B of LDS per root: at budget 5000 that is
//              40 448 B, FOUR roots per CU instead of three -- the 1024-root shard of BASELINE C4 stays on the
//              low-latency variant.
// opd_wide_kernel (below): the bounds array lives in HBM/L2 (class-contiguous, so a class re-scan is one coalesced read)
//              and LDS only holds the window of the closing pass: 8 waves per SIMD instead of 3-4 per CU.  A wave of this
//              planner is ONE chain of ~300 dependent instructions per expansion (two cross-lane argmaxes of ~345 cycles,
//              an LDS re-scan, two round trips that hide under them): ten times more resident roots fill the stalls and
//              multiply the batch throughput.
// NONNEG: every finite bound is >= +0.0 (gamma in [0, 1), terminal reward >= 0; rewards are range-checked): the reductions
//              take the zero-fill DPP steps of wave.hpp.  What a lone wave's expansion costs, phase by phase, and what was
//              tried on it: profiles/r03_opd_pipelining.md.
// the lowest id (>= 0) among the lanes of a predicate: the largest INT_MAX - id, zero-fill DPP steps
__device__ __forceinline__ int lowest_id(bool in, int id)
{
    int key = in ? 0x7fffffff - id : 0;
    key = imax_step_zero<0x111>(key); key = imax_step_zero<0x112>(key); key = imax_step_zero<0x114>(key);
    key = imax_step_zero<0x118>(key); key = imax_step_zero<0x142>(key); key = imax_step_zero<0x143>(key);
    return 0x7fffffff - __builtin_amdgcn_readlane(key, 63);
}

// ---- the leaf to expand (deterministic.py:110: the first maximal upper bound), from the per-lane class bests.
// The maximum M found by a full selection stays THE maximum for as long as some class best still equals it: no bound above
// it can appear without being noticed (a re-scan returns a leaf that was already there; the callers compare new children
// with M and clear `ok`).  OPD's trees are full of exact ties -- on the benchmark tables nine selections in ten pick among
// leaves that share the bound of the previous one -- so the usual selection is only the third of the three reductions of
// wave.hpp's key argmax: the lowest id among the lanes that hold M (and not even that when one lane holds it).
template <bool NONNEG>
__device__ __forceinline__ int select_drain(double cbu, int cbid, int lane, bool &ok, int &m_hi, int &m_lo)
{
    bool in = ok && __double2hiint(cbu) == m_hi && __double2loint(cbu) == m_lo;
    unsigned long long cand = ballot64(in);
    if (cand == 0ull) { // (wave-uniform) a new maximum: the first two reductions
        if (NONNEG) { // (bounds >= +0.0 or -inf: the bit pattern is the key, zero-fill DPP steps -- wave.hpp)
            const int hi = __double2hiint(cbu);
            const unsigned lo = (unsigned)__double2loint(cbu);
            int kh = hi < 0 ? 0 : hi;
            kh = imax_step_zero<0x111>(kh); kh = imax_step_zero<0x112>(kh); kh = imax_step_zero<0x114>(kh);
            kh = imax_step_zero<0x118>(kh); kh = imax_step_zero<0x142>(kh); kh = imax_step_zero<0x143>(kh);
            const int mh = __builtin_amdgcn_readlane(kh, 63);
            const bool c1 = hi == mh;
            unsigned l1 = c1 ? lo : 0u;
            l1 = umax_step_zero<0x111>(l1); l1 = umax_step_zero<0x112>(l1); l1 = umax_step_zero<0x114>(l1);
            l1 = umax_step_zero<0x118>(l1); l1 = umax_step_zero<0x142>(l1); l1 = umax_step_zero<0x143>(l1);
            const unsigned ml = (unsigned)__builtin_amdgcn_readlane((int)l1, 63);
            in = c1 && lo == ml;
            m_hi = mh; m_lo = (int)ml;
        } else {
            const double uc = cbu + 0.0;
            const int hi = __double2hiint(uc), sg = hi >> 31;
            const int kh = hi ^ (sg & 0x7fffffff);
            const unsigned kl = (unsigned)(__double2loint(uc) ^ sg);
            int mh = kh;
            MP_DPP_REDUCE_WAVE("v_max_i32_dpp", mh);
            mh = __builtin_amdgcn_readlane(mh, 63);
            const bool c1 = kh == mh;
            unsigned l1 = c1 ? kl : 0u;
            MP_DPP_REDUCE_WAVE("v_max_u32_dpp", l1);
            const unsigned ml = (unsigned)__builtin_amdgcn_readlane((int)l1, 63);
            in = c1 && kl == ml;
            const int ms = mh >> 31;
            m_hi = mh ^ (ms & 0x7fffffff); m_lo = (int)(ml ^ (unsigned)ms);
        }
        cand = ballot64(in);
        ok = true;
    }
    if (__popcll(cand) == 1) return __builtin_amdgcn_readlane(cbid, __ffsll((long long)cand) - 1);
    return lowest_id(in, cbid);
}

// ---- the best leaf of a re-scanned row from the per-lane bests (ru, rid).  Nothing exceeds the maximum M that select_drain
// holds, so if some lane's best EQUALS M the row's best is M and only the lowest id among those lanes is wanted (with the
// sibling layout: 85 % of the re-scans on the benchmark tables -- siblings tie); else the full key argmax.
template <bool NONNEG>
__device__ __forceinline__ void rescan_best(double &ru, int &rid, int lane, int m_hi, int m_lo)
{
    const bool in = __double2hiint(ru) == m_hi && __double2loint(ru) == m_lo;
    const unsigned long long eq = ballot64(in);
    if (eq != 0ull) { // (wave-uniform)
        rid = __popcll(eq) == 1 ? __builtin_amdgcn_readlane(rid, __ffsll((long long)eq) - 1) : lowest_id(in, rid);
        ru = __hiloint2double(m_hi, m_lo);
        return;
    }
    if (NONNEG) wave_argmax_keys_nonneg(ru, rid); else wave_argmax_keys(ru, rid);
}

template <bool EXPG, bool NONNEG>
__global__ __launch_bounds__(64) void opd_kernel(OpdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int T = p.T;                                                // odd, >= ceil(cap / 64)
    double *leafU = lds;                                                        // [64 * T]
    int32_t *exp_lds = EXPG ? p.expanded + (long)blockIdx.x * (p.K > 0 ? p.K : 1) // parent map in HBM
                            : reinterpret_cast<int32_t *>(lds + 64 * T);        // [K]
#define LU(id) leafU[((id) & 63) * T + ((id) >> 6)]
    const int lane = threadIdx.x;
    const int root = blockIdx.x;
    const int A = p.A;
    const long base = (long)root * p.cap;
    OpdNode *NA = reinterpret_cast<OpdNode *>(p.L) + base;
    double *U = p.U + base, *RW = p.reward + base;
    constexpr int32_t DONE_FLAG = 1 << 30; // in OpdNode::depth
    const uint32_t done_bit = p.done_on_next ? 2u : 1u;
    const double ninf = -INFINITY;

    // deterministic.py:10-19 root: L = U = 0, depth 0
    if (lane == 0) {
        OpdNode n0;
        n0.L = 0.0; n0.state = p.root_state[root]; n0.depth = 0;
        NA[0] = n0;
        RW[0] = 0.0;
        LU(0) = 0.0;
    }
    __syncthreads();
    int n_nodes = 1;
    // children of available actions = planner.step calls (deterministic.py:41), counted per lane
    int real_mine = 0;
    int status = MP_OK;
    int k_done = 0;
    // best leaf of this lane's class (ids == lane mod 64); -inf / INT_MAX when the class has no leaf
    double cbu = lane == 0 ? 0.0 : ninf;
    int cbid = lane == 0 ? 0 : 0x7fffffff;

#ifdef MP_PROFILE
    long long t_scan = 0, t_exp = 0, t_all0 = clock64(), t_f[5] = {0, 0, 0, 0, 0};
    long long t_p[6] = {0, 0, 0, 0, 0, 0};
    int n_follow = 0, n_recent = 0, n_tie_top = 0;
#define PROF_T(x) const long long x = clock64()
#define ANCHOR(v) { int t__; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(t__) : "v"(v) : "memory"); }
#else
#define PROF_T(x)
#endif
    for (int k = 0; k < p.K; ++k) {
        PROF_T(c0);
        // ---- deterministic.py:110: first maximal upper bound among the leaves
        // (the full reduction every time: select_drain's shortcuts are wave-uniform branches, and a LONE wave pays more for a
        // readlane -> compare -> branch than for the twelve steps they save -- profiles/r05_opd_wide.md)
        double bu = cbu;
        int leaf = cbid;
        if (NONNEG) wave_argmax_nonneg(bu, leaf); else wave_argmax(bu, leaf);
        const int cls = leaf & 63;
#ifdef MP_PROFILE2
        ANCHOR(leaf); const long long pa = clock64();
        if (leaf >= n_nodes - A) ++n_follow;
        if (leaf >= n_nodes - 8 * A) ++n_recent;
#endif
        // the leaf's record is needed by the expansion only: fetch it now, under the class re-scan
        const uint4 leaf_raw = *reinterpret_cast<const uint4 *>(&NA[leaf]);
        // the selected leaf stops being one; re-derive the best leaf of its class from LDS
        if (lane == 0) LU(leaf) = ninf;
        // A workgroup is ONE wavefront: its LDS operations execute in program order, so lane 0's store is seen by the
        // other lanes' later reads without a barrier.  Only the compiler is ordered here -- a full __syncthreads() would
        // also wait for the record fetch above.
        __builtin_amdgcn_wave_barrier();
        OpdNode pn; // (one dwordx4: the compiler splits the struct load when a field is read first)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 rc_raw;
        unsigned long long g1_bits, gdiv_bits, tdiv_bits;
        int d;
        {
            const double *row = leafU + cls * T;
            const int cnt = (n_nodes - cls + 63) >> 6; // ids cls, cls + 64, ... < n_nodes
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
            // The leaf's record has had the scan's LDS round trip to arrive: request the model records of its |A| actions
            // NOW, so that they fly under the reduction below (every lane loads -- lanes >= |A| the last action's record,
            // unused: a load under a divergent branch would be waited for where the branch closes).
            pn.L = __hiloint2double((int)leaf_raw.y, (int)leaf_raw.x); pn.state = (int32_t)leaf_raw.z; pn.depth = (int32_t)leaf_raw.w;
            // The loads are inline assembly: the compiler sinks a plain load to its first use, below the reduction, and its
            // s_waitcnt bookkeeping does not see these -- they are waited for by hand where the children are computed.
            {
                const int cj = (lane - n_nodes) & 63; // child j is computed by lane (g + j) mod 64, the owner of its class
                const Rec *src = p.rec + ((long)pn.state * A + (cj < A ? cj : A - 1));
                // (ru passes through: the reduction below cannot be scheduled above the request)
                asm volatile("global_load_dwordx4 %0, %2, off" : "=v"(rc_raw), "+v"(ru) : "v"(src) : "memory");
            }
            // wave-uniform depth in an SGPR: the three gamma tables come through the scalar cache, not the TA
            d = __builtin_amdgcn_readfirstlane((pn.depth & (DONE_FLAG - 1)) + 1);
            {
                const double *s0 = p.g1 + d, *s1 = p.gdiv + d, *s2 = p.tdiv + d;
                asm volatile("s_load_dwordx2 %0, %4, 0x0\n\ts_load_dwordx2 %1, %5, 0x0\n\ts_load_dwordx2 %2, %6, 0x0"
                             : "=&s"(g1_bits), "=&s"(gdiv_bits), "=&s"(tdiv_bits), "+v"(rid) : "s"(s0), "s"(s1), "s"(s2) : "memory");
            }
#ifdef MP_PROFILE2
            ANCHOR(__double2hiint(ru)); const long long pb = clock64();
            t_p[0] += pa - c0; t_p[1] += pb - pa;
#endif
            if (NONNEG) wave_argmax_nonneg(ru, rid); else wave_argmax(ru, rid);
            if (lane == cls) { cbu = ru; cbid = rid; }
        }
#ifdef MP_PROFILE2
        ANCHOR(cbid);
#endif
        PROF_T(c1);
        // ---- DeterministicNode.expand, deterministic.py:28-43
        const int g = n_nodes; // first child
#ifdef MP_PROFILE2
        ANCHOR((int)leaf_raw.w); const long long pd = clock64();
        t_p[2] += pd - c1;
#endif
        // (cbid passes through: the wait cannot be scheduled above the reduction)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(rc_raw), "+s"(g1_bits), "+s"(gdiv_bits), "+s"(tdiv_bits), "+v"(cbid) : : "memory");
        const double g1d = __longlong_as_double((long long)g1_bits), gdivd = __longlong_as_double((long long)gdiv_bits),
                     tdivd = __longlong_as_double((long long)tdiv_bits);
        Rec rc;
        rc.next = (int32_t)rc_raw.x; rc.flags = rc_raw.y; rc.reward = __hiloint2double((int)rc_raw.w, (int)rc_raw.z);
        bool bad = false, avail = false;
        double Uc_mine = ninf;
        const int cj = (lane - g) & 63;
        if (cj < A) {
            const double r = rc.reward;
            // deterministic.py:32-35: only the actions state.get_available_actions() lists get a child.  The slot of
            // an unavailable action stays in the id space (ids advance by |A| per expansion in every root) as a
            // PHANTOM with L = U = -inf: never the best leaf, never a maximum of a backup, never a tie of the plan;
            // the tree export drops it.
            avail = (rc.flags & 4u) != 0;
            bad = avail && (!(0.0 <= r) || !(r <= 1.0)); // deterministic.py:46-47
            const bool dn = (rc.flags & done_bit) != 0;
            // deterministic.py:45-65 update()
            double Lc = pn.L + g1d * r;
            double Uc = Lc + gdivd;
            if (dn) {
                const double nv = Lc + tdivd;
                Lc = nv; Uc = nv;
            }
            if (!avail) { Lc = ninf; Uc = ninf; }
            const int c = g + cj;
            OpdNode cn;
            cn.L = Lc; cn.state = rc.next; cn.depth = d | (dn ? DONE_FLAG : 0);
            NA[c] = cn;
            RW[c] = r;
            LU(c) = Uc;
            Uc_mine = Uc;
        }
#ifdef MP_PROFILE2
        ANCHOR(__double2hiint(Uc_mine)); const long long pe = clock64();
        t_p[3] += pe - pd;
#endif
        if (lane == 0) {
            exp_lds[k] = leaf;
        }
        n_nodes += A;
        real_mine += avail ? 1 : 0;
        k_done = k + 1;
        if (ballot64(bad) != 0ull) { status = MP_ERR_REWARD_RANGE; break; }
        __builtin_amdgcn_wave_barrier();
        // the (at most one, |A| <= 64) new child that falls in this lane's class may beat its cached
        // best; on equality the older (lower id) leaf stays, as in the reference's list order
        if (Uc_mine > cbu) { cbu = Uc_mine; cbid = g + cj; } // (-inf in the lanes without a child)
#ifdef MP_PROFILE
        { ANCHOR(__double2hiint(cbu)); const long long c2 = clock64(); t_scan += c1 - c0; t_exp += c2 - c1;
#ifdef MP_PROFILE2
          t_p[4] += c2 - pe;
#endif
        }
#endif
    }
    PROF_T(cf0);
    __syncthreads();

    if (status == MP_OK) {
        // ---- all backup_to_root calls at once (deterministic.py:67-79): bottom-up max.
        // Upper bounds: an internal node's U is the max leaf U of its subtree, so the root's is the max over
        // all leaves -- one parallel reduction; the other internal U's are only ever read by a tree export,
        // which fills them in on the host from the leaf values stored here (-inf marks an internal node).
        double root_upper = ninf;
        for (int i = lane; i < n_nodes; i += 64) {
            const double u = LU(i);
            U[i] = u;
            if (u > root_upper) root_upper = u;
        }
        root_upper = wave_max(root_upper);
        __syncthreads();
        PROF_T(cf1);
        Pcg64 gen;
        gen.load(p.rng + (long)root * 6);
        int len = 0;
        double root_lower;
        if (!p.closing_chain && closing_compact_fits(p.K, A, p.cap, 64L * T * 8)) {
            // pointer jumping over the expansion tree + a prepared plan walk (opd_closing.hpp)
            len = closing_compact(lds, p.K, k_done, n_nodes, A, exp_lds, [&](int id) { return NA[id].L; },
                                  [&](int id, double v) { NA[id].L = v; }, gen,
                                  p.plans ? p.plans + (long)root * p.max_plan_len : nullptr, p.max_plan_len, root_lower);
            PROF_T(cf2); PROF_T(cf3); PROF_T(cf4);
#ifdef MP_PROFILE
            t_f[0] = cf1 - cf0; t_f[1] = cf2 - cf1; t_f[2] = cf3 - cf2; t_f[3] = cf4 - cf3; t_f[4] = cf4;
#endif
        } else {
            // lower bounds: same pass over the creation-time L values
            for (int i = lane; i < n_nodes; i += 64) LU(i) = NA[i].L;
            __syncthreads();
            PROF_T(cf2);
            // (|A| < 4: the tables of opd_closing.hpp do not fit beside each other.)  The backups as a fixed point on the
            // node array: L[parent of expansion k] = max over the children group k; children are expanded after their
            // parents, so the 64-expansion chunks are taken from the last one down, one expansion per lane, and each
            // chunk is repeated until none of its lanes computed a new maximum.
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
                    __builtin_amdgcn_wave_barrier(); // (one wavefront: LDS operations execute in program order)
                    if (!any64(changed)) break;
                }
            }
            __syncthreads();
            PROF_T(cf3);
            for (int k = lane; k < k_done; k += 64) {
                const int n = exp_lds[k];
                NA[n].L = LU(n);
            }
            PROF_T(cf4);
#ifdef MP_PROFILE
            t_f[0] = cf1 - cf0; t_f[1] = cf2 - cf1; t_f[2] = cf3 - cf2; t_f[3] = cf4 - cf3; t_f[4] = cf4;
#endif
            // ---- get_plan (abstract.py:143-156) with DeterministicNode.selection_rule
            // (deterministic.py:21-26): random_argmax over the children's lower bounds.
            // A node's children are group 1 + k*A where k is its expansion index.  The bounds array now becomes the
            // node -> expansion-index map (a negative quiet NaN with payload k in the slot of every expanded node), so
            // that a level of the descent is ONE round trip -- the children's final lower bounds from their records,
            // their slots from LDS -- instead of a search of the parent map.
            __syncthreads(); // the final lower bounds are in the records (s_waitcnt vmcnt(0))
            root_lower = LU(0);
            __builtin_amdgcn_wave_barrier();
            for (int k = lane; k < k_done; k += 64) LU(exp_lds[k]) = __hiloint2double((int)0xFFF80000, k);
            __syncthreads();
            int kcur = k_done > 0 ? 0 : -1; // the first expansion is always the root
            while (kcur >= 0) {
                const int fc = 1 + kcur * A;
                const double l = lane < A ? NA[fc + lane].L : ninf;
                const double slot = lane < A ? LU(fc + lane) : 0.0;
                const double m = A <= 16 ? row0_max(l) : wave_max(l);
                const unsigned long long ties = ballot64(lane < A && l == m);
                const int nt = __popcll(ties);
                int pick = (int)gen.below((uint32_t)nt); // uniform across lanes (same state, same draws)
                unsigned long long t = ties;
                while (pick-- > 0) t &= t - 1;
                const int a = __ffsll((long long)t) - 1;
                if (lane == 0 && p.plans && len < p.max_plan_len) p.plans[(long)root * p.max_plan_len + len] = a;
                ++len;
                const int shi = __builtin_amdgcn_readlane(__double2hiint(slot), a), slo = __builtin_amdgcn_readlane(__double2loint(slot), a);
                kcur = ((unsigned)shi == 0xFFF80000u) ? slo : -1; // expanded: its k; a leaf: the plan ends
            }
        }
#ifdef MP_PROFILE
        t_f[4] = clock64() - t_f[4];
#endif
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
#ifdef MP_PROFILE
    if (root == 0 && lane == 0)
        printf("opd prof root0: K=%d total=%lld scan=%lld expand=%lld final=%lld [U %lld  Lload %lld  backward %lld  Lstore %lld  "
               "descent %lld] (clock64 ticks)\n", p.K, (long long)(clock64() - t_all0), t_scan, t_exp, (long long)(clock64() - cf0),
               t_f[0], t_f[1], t_f[2], t_f[3], t_f[4]);
#ifdef MP_PROFILE2
    if (root == 0 && lane == 0)
        printf("opd prof2 root0: argmax %lld  rescan-reads %lld  rescan-argmax %lld  leaf-wait+tables %lld  model+children %lld  tail %lld\n",
               t_p[0], t_p[1], t_scan - t_p[0] - t_p[1], t_p[2], t_p[3], t_p[4]);
    if (root < 4 && lane == 0) printf("opd prof2 root%d: leaf among the previous expansion's children %d, among the last 8 expansions' %d, ties at the top argmax %d of %d\n", root, n_follow, n_recent, n_tie_top, k_done);
#endif
#endif
    int n_real = real_mine;
    for (int off = 32; off > 0; off >>= 1) n_real += __shfl_xor(n_real, off);
    if (lane == 0) {
        if (p.status) p.status[root] = status;
        if (p.env_steps) p.env_steps[root] = (int64_t)n_real;
        p.n_nodes_out[root] = n_nodes;
    }
    if (EXPG) {
        for (int k = k_done + lane; k < p.K; k += 64) p.expanded[(long)root * p.K + k] = -1;
    } else {
        for (int k = lane; k < p.K; k += 64) p.expanded[(long)root * p.K + k] = k < k_done ? exp_lds[k] : -1;
    }
#undef LU
}


// ---- the high-occupancy variant as its own kernel (8 waves per SIMD; 8192 roots of BASELINE C4 and beyond).
// At that occupancy the chip is busy, not waiting: VALU issue 68-72 % of all SIMD cycles (profiles/r02_opd_units.md), so
// what counts is the number of instructions an expansion issues:
//   * the argmaxes use the key form (wave.hpp: 18 single-instruction DPP steps);
//   * the selected leaf's slot of the bounds array gets a negative quiet NaN whose payload is the expansion index k
//     instead of -inf: a NaN never compares greater, so it is as dead to every selection as -inf, and it makes the bounds
//     array the node -> k map.  The parent map (k -> node) is then not stored per expansion at all -- the closing pass
//     over the bounds array, which exists anyway, scatters it -- and the plan descent reads a child's k from its slot
//     (one round trip per level) instead of searching the map.
// Tried and dropped here: one 8-byte store instruction for everything 8 bytes wide (children's bounds, the leaf's slot,
// the children's rewards from lanes |A|+1..2|A| that gather the same records) plus a pair-wise dwordx4 class re-scan that
// skips the leaf in registers -- five vector-memory instructions per expansion instead of nine, but ten more VALU
// instructions for the lane roles: 3.66 ms against 3.49 at 8192 roots.  TA_BUSY did not move (74 %): it counts a unit
// with requests in flight, not a unit out of issue slots.
template <bool NONNEG, bool SMALLT, bool SIB>
__global__ __launch_bounds__(64, 8) void opd_wide_kernel(OpdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int T = SIB ? p.Tsib : p.T;
    const int lgP = p.lgP, P = 1 << lgP; // SIB: a leaf's code is (group << lgP) | child index
    double *leafU = p.leaf_global + (long)blockIdx.x * 64 * T;
#define LU(id) leafU[((id) & 63) * T + ((id) >> 6)]
    const int lane = threadIdx.x, root = blockIdx.x, A = p.A;
    const long base = (long)root * p.cap;
    OpdNode *NA = reinterpret_cast<OpdNode *>(p.L) + base;
    double *RW = p.reward + base;
    constexpr int32_t DONE_FLAG = 1 << 30;
    const uint32_t done_bit = p.done_on_next ? 2u : 1u;
    const double ninf = -INFINITY;

    if (lane == 0) {
        OpdNode n0;
        n0.L = 0.0; n0.state = p.root_state[root]; n0.depth = 0;
        NA[0] = n0;
        RW[0] = 0.0;
        if (!SIB) LU(0) = 0.0;
    }
    if (SIB && lane < A) leafU[lane] = lane == 0 ? 0.0 : ninf; // group 0 of row 0: the root (its other slots are never leaves)
    __syncthreads();
    int n_nodes = 1, real_mine = 0, status = MP_OK, k_done = 0;
    double cbu = lane == 0 ? 0.0 : ninf;
    int cbid0 = lane == 0 ? 0 : 0x7fffffff; // (SIB: the root's code is 0 as well)

    bool drain_ok = false;
    int drain_hi = 0, drain_lo = 0;
    if constexpr (SIB) {
    // ---- sibling layout.  The bounds array has 64 rows; the |A| children of expansion k sit TOGETHER, in group
    // e = k + 1 (group 0 holds the root): row e mod 64, slots (e / 64) * |A| .. + |A| - 1, so an expansion writes its
    // children's bounds as ONE request of |A| * 8 contiguous bytes where the residue-class layout scattered
    // them over |A| rows -- this kernel waits on the L1's outstanding requests three quarters of the time, most of them
    // partial-line writes (profiles/r05_opd_wide.md).  A leaf is named by its code e * P + j, ordered like the ids.
    // Lane l caches the best leaf of row l; the children of an expansion all belong to ONE row.
    const int cb0 = (((lane / A) << 6) << lgP) | (lane % A);                   // code of slot t = lane of row 0 ...
    const int cb1 = ((((lane + 64) / A) << 6) << lgP) | ((lane + 64) % A);     // ... and of slot lane + 64
    int cbid = cbid0;
    for (int k = 0; k < p.K; ++k) {
        // ---- deterministic.py:110: first maximal upper bound among the leaves (see the residue-class loop below)
        const int leaf = select_drain<NONNEG>(cbu, cbid, lane, drain_ok, drain_hi, drain_lo);
        const int leaf_s = __builtin_amdgcn_readfirstlane(leaf);
        const int el = leaf_s >> lgP, jl = leaf_s & (P - 1);
        const int cls = el & 63;
        const int leaf_id = el == 0 ? 0 : 1 + (el - 1) * A + jl;
        const double *row = leafU + cls * T;
        // the selected leaf stops being one: its slot becomes the node -> expansion-index map entry (one lane, one request)
        if (lane == 0) leafU[cls * T + (el >> 6) * A + jl] = __hiloint2double((int)0xFFF80000, k);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // (a wavefront's vector-memory operations are performed in order)
        const int cnt = (((k - cls) >> 6) + 1) * A; // groups e <= k of this row
        double u0 = ninf, u1 = ninf;
        typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
        u32x4 lr;
        const OpdNode *lp = NA + leaf_id;
        if (SMALLT) {
            u0 = row[lane < cnt ? lane : 0];
            if (cnt > 64) {
                u1 = row[lane + 64 < cnt ? lane + 64 : 0];
                OPD_LEAF_SLOAD(3, lr, lp);
            } else {
                OPD_LEAF_SLOAD(2, lr, lp);
            }
        } else {
            OPD_LEAF_SLOAD(1, lr, lp);
        }
        OpdNode pn;
        pn.L = __hiloint2double((int)lr.y, (int)lr.x); pn.state = (int32_t)lr.z; pn.depth = (int32_t)lr.w;
        const bool mine = lane < A;
        Rec rc;
        rc.next = 0; rc.flags = 0; rc.reward = 0.0;
        if (mine) rc = (p.rec + (long)pn.state * A)[lane];
        {
            double ru = ninf;
            int rid = 0x7fffffff;
            const int cshift = cls << lgP;
            if (SMALLT) {
                if (lane < cnt && u0 > ru) { ru = u0; rid = cb0 | cshift; }
                if (cnt > 64 && lane + 64 < cnt && u1 > ru) { ru = u1; rid = cb1 | cshift; }
            } else {
                for (int t = lane; t < cnt; t += 64) {
                    const double v0 = row[t];
                    if (v0 > ru) { ru = v0; rid = ((((t / A) << 6) << lgP) | (t % A)) | cshift; }
                }
            }
            rescan_best<NONNEG>(ru, rid, lane, drain_hi, drain_lo);
            if (lane == cls) { cbu = ru; cbid = rid; }
        }
        const int d = __builtin_amdgcn_readfirstlane((pn.depth & (DONE_FLAG - 1)) + 1);
        typedef const double __attribute__((address_space(4))) *scalar_f64;
        const double g1d = ((scalar_f64)(unsigned long long)p.g1)[d], gdivd = ((scalar_f64)(unsigned long long)p.gdiv)[d],
                     tdivd = ((scalar_f64)(unsigned long long)p.tdiv)[d];
        double r = rc.reward;
        asm volatile("" : "+v"(r));
        const bool avail = mine && (rc.flags & 4u) != 0;    // deterministic.py:32-35 (phantom slots: see opd_kernel)
        const bool bad = avail && (!(0.0 <= r) || !(r <= 1.0)); // deterministic.py:46-47
        const bool dn = (rc.flags & done_bit) != 0;
        double Lc = pn.L + g1d * r;                       // deterministic.py:45-65 update()
        double Uc = Lc + gdivd;
        if (dn) {
            const double nv = Lc + tdivd;
            Lc = nv; Uc = nv;
        }
        if (!avail) { Lc = ninf; Uc = ninf; }
        const int e = k + 1, re = e & 63;
        if (mine) {
            const int c = n_nodes + lane;
            OpdNode cn;
            cn.L = Lc; cn.state = rc.next; cn.depth = d | (dn ? DONE_FLAG : 0);
            NA[c] = cn;
            RW[c] = r;
        }
        if (mine) leafU[re * T + (e >> 6) * A + lane] = Uc; // the whole group in one request
        n_nodes += A;
        real_mine += avail ? 1 : 0;
        k_done = k + 1;
        if (ballot64(bad) != 0ull) { status = MP_ERR_REWARD_RANGE; break; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the next re-scan may read these children through memory, from other lanes
        // the children against the best of THEIR row (lane re)
        const double cbu_re = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cbu), re), __builtin_amdgcn_readlane(__double2loint(cbu), re));
        if (ballot64(mine && Uc > cbu_re) != 0ull) { // (wave-uniform)
            const double um = mine ? Uc : ninf;
            const double m = A <= 16 ? row0_max(um) : wave_max(um);
            const int jm = __ffsll((long long)ballot64(mine && Uc == m)) - 1; // lowest id among equal bounds
            if (lane == re) { cbu = m; cbid = (e << lgP) + jm; }
            // (in exact arithmetic a child's bound never exceeds its parent's; rounded, gamma^(d-1) r + gamma^d / (1 - gamma)
            // can land an ulp above gamma^(d-1) / (1 - gamma))
            if (m > __hiloint2double(drain_hi, drain_lo)) drain_ok = false;
        }
    }
    cbid0 = cbid;
    } else {
    int cbid = cbid0;
    for (int k = 0; k < p.K; ++k) {
        // ---- deterministic.py:110: first maximal upper bound among the leaves
        const int leaf = select_drain<NONNEG>(cbu, cbid, lane, drain_ok, drain_hi, drain_lo);
        const int leaf_s = __builtin_amdgcn_readfirstlane(leaf);
        const int cls = leaf_s & 63;
        // the selected leaf stops being one: its slot becomes the node -> expansion-index map entry (one lane, one request)
        if (lane == 0) LU(leaf_s) = __hiloint2double((int)0xFFF80000, k);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // (a wavefront's vector-memory operations are performed in order)
        const double *row = leafU + cls * T;
        const int cnt = (n_nodes - cls + 63) >> 6;
        double u0 = ninf, u1 = ninf;
        // The leaf's record is wave-uniform: a SCALAR load (no address-coalescer time, results in scalar registers), past the
        // scalar cache (glc: the record was written by this wave's vector stores, the scalar cache does not see those).  The
        // stores that wrote it are the previous expansion's at the latest: vmcnt(N), N = the vector-memory operations of THIS
        // expansion issued so far, waits for exactly those.
        typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
        u32x4 lr;
        const OpdNode *lp = NA + leaf_s;
        if (SMALLT) {
            u0 = row[lane < cnt ? lane : 0];
            if (cnt > 64) {
                u1 = row[lane + 64 < cnt ? lane + 64 : 0];
                OPD_LEAF_SLOAD(3, lr, lp);
            } else {
                OPD_LEAF_SLOAD(2, lr, lp);
            }
        } else {
            OPD_LEAF_SLOAD(1, lr, lp);
        }
        OpdNode pn;
        pn.L = __hiloint2double((int)lr.y, (int)lr.x); pn.state = (int32_t)lr.z; pn.depth = (int32_t)lr.w;
        const int g = n_nodes;
        const int j = (lane - g) & 63; // child j is computed by lane (g + j) mod 64, the owner of its class
        const bool mine = j < A;
        Rec rc;
        rc.next = 0; rc.flags = 0; rc.reward = 0.0;
        if (mine) rc = (p.rec + (long)pn.state * A)[j];
        {
            double ru = ninf;
            int rid = 0x7fffffff;
            if (SMALLT) {
                if (lane < cnt && u0 > ru) { ru = u0; rid = cls + (lane << 6); }
                if (cnt > 64 && lane + 64 < cnt && u1 > ru) { ru = u1; rid = cls + ((lane + 64) << 6); }
            } else {
                for (int t = lane; t < cnt; t += 128) { // two reads in flight per trip
                    const double v0 = row[t];
                    const double v1 = t + 64 < cnt ? row[t + 64] : ninf;
                    if (v0 > ru) { ru = v0; rid = cls + (t << 6); }
                    if (v1 > ru) { ru = v1; rid = cls + ((t + 64) << 6); }
                }
            }
            rescan_best<NONNEG>(ru, rid, lane, drain_hi, drain_lo);
            if (lane == cls) { cbu = ru; cbid = rid; }
        }
        const int d = __builtin_amdgcn_readfirstlane((pn.depth & (DONE_FLAG - 1)) + 1);
        typedef const double __attribute__((address_space(4))) *scalar_f64;
        const double g1d = ((scalar_f64)(unsigned long long)p.g1)[d], gdivd = ((scalar_f64)(unsigned long long)p.gdiv)[d],
                     tdivd = ((scalar_f64)(unsigned long long)p.tdiv)[d];
        // (computed by every lane -- lanes >= |A| on a copy of the last record, discarded -- so that the wait for the
        // records sits in straight-line code and only the three stores are conditional)
        double r = rc.reward;
        asm volatile("" : "+v"(r));
        const bool avail = mine && (rc.flags & 4u) != 0;    // deterministic.py:32-35 (phantom slots: see opd_kernel)
        const bool bad = avail && (!(0.0 <= r) || !(r <= 1.0)); // deterministic.py:46-47
        const bool dn = (rc.flags & done_bit) != 0;
        double Lc = pn.L + g1d * r;                       // deterministic.py:45-65 update()
        double Uc = Lc + gdivd;
        if (dn) {
            const double nv = Lc + tdivd;
            Lc = nv; Uc = nv;
        }
        if (!avail) { Lc = ninf; Uc = ninf; }
        const double Uc_mine = Uc;
        const int c = g + j;
        if (mine) {
            OpdNode cn;
            cn.L = Lc; cn.state = rc.next; cn.depth = d | (dn ? DONE_FLAG : 0);
            NA[c] = cn;
            RW[c] = r;
            leafU[__umul24(lane, T) + (c >> 6)] = Uc;     // = LU(c): c mod 64 is this lane
        }
        n_nodes += A;
        real_mine += avail ? 1 : 0;
        k_done = k + 1;
        if (ballot64(bad) != 0ull) { status = MP_ERR_REWARD_RANGE; break; }
        // (in exact arithmetic a child's bound never exceeds its parent's; rounded, gamma^(d-1) r + gamma^d / (1 - gamma) can
        // land an ulp above gamma^(d-1) / (1 - gamma))
        if (ballot64(mine && Uc_mine > __hiloint2double(drain_hi, drain_lo)) != 0ull) drain_ok = false;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the next re-scan may read these children through memory, from other lanes
        if (mine && Uc_mine > cbu) { cbu = Uc_mine; cbid = c; }
    }
    cbid0 = cbid;
    }
    __syncthreads();

    // Everything only the closing passes need is read from the kernel-argument segment HERE: loaded at entry (as the
    // compiler does with by-value arguments) those twelve pointers sit in SGPRs through the main loop, which at the 80
    // SGPRs of 8 waves per SIMD meant 17 spill reloads per expansion (a tenth of its VALU instructions).
    const OpdArgs __attribute__((address_space(4))) *q;
    {
        unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka)); // the loads below cannot move above this point
        q = (const OpdArgs __attribute__((address_space(4))) *)ka;
    }
    double *U = q->U + base;
    int32_t *exp_map = q->expanded + (long)root * (q->K > 0 ? q->K : 1);
    int32_t *const plans = q->plans, *const plan_len = q->plan_len;
    const int max_plan_len = q->max_plan_len;

    // closing pass over the bounds array: leaf upper bounds out, the root's upper bound, and the parent map scattered from
    // the NaN payloads (every expanded node carries its k)
    double root_upper = ninf;
    if constexpr (SIB) {
        // (four rows per trip: up to eight independent reads in flight; the slots of a row that no group has reached yet
        // are skipped by index, whatever they hold)
        const int tg0 = lane / A, tj0 = lane % A, tg1 = (lane + 64) / A, tj1 = (lane + 64) % A;
        auto emit = [&](int r, int t, int cnt, double u, int tg, int jj) { // slot t = group tg of its row, child jj
            const int e = (tg << 6) | r;
            if (t < cnt && (e > 0 || jj == 0)) {
                const int id = e == 0 ? 0 : 1 + (e - 1) * A + jj;
                U[id] = u != u ? ninf : u;
                if (u > root_upper) root_upper = u;
                if (u != u) exp_map[__double2loint(u)] = id;
            }
        };
        for (int r0 = 0; r0 < 64 && r0 <= k_done; r0 += 4) {
            int cnt[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) cnt[a] = r0 + a <= k_done ? (((k_done - r0 - a) >> 6) + 1) * A : 0;
            if (SMALLT) {
                double uu[4][2];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    uu[a][0] = leafU[(r0 + a) * T + lane];
                    uu[a][1] = leafU[(r0 + a) * T + (lane + 64 < T ? lane + 64 : lane)];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    emit(r0 + a, lane, cnt[a], uu[a][0], tg0, tj0);
                    emit(r0 + a, lane + 64, cnt[a], uu[a][1], tg1, tj1);
                }
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    for (int t = lane; t < cnt[a]; t += 64) emit(r0 + a, t, cnt[a], leafU[(r0 + a) * T + t], t / A, t % A);
            }
        }
    } else {
        for (int i = lane; i < n_nodes; i += 64) {
            const double u = LU(i);
            U[i] = u != u ? ninf : u; // (-inf marks an expanded node for the export, as in the other variants)
            if (u > root_upper) root_upper = u;
            if (u != u) exp_map[__double2loint(u)] = i;
        }
    }
    root_upper = wave_max(root_upper);
    __syncthreads();

    if (status == MP_OK) {
        // Lower bounds without the bounds array: the creation-time L of every node sits in its record, and the reverse
        // sweep only ever READS the |A| children of the step at hand -- a window of ids that slides down by |A| per
        // step -- so `chunk` steps are served from an LDS window of chunk * |A| values (2.5 KB at |A| = 5), filled by
        // one coalesced read of the records.  A step is LDS read -> DPP max -> LDS write (when the parent is inside the
        // window) + one fire-and-forget 8-byte store of the final L into the parent's record; the stores are waited
        // for once per window, before the next fill (parents below the window are read back then).  Through the L2
        // every step was two dependent round trips (3 400 cycles at 8 waves per SIMD; 37 % of the kernel).
        double *win = lds;
        const int C = q->chunk;
        for (int k0 = k_done - 1; k0 >= 0;) {
            const int kb = k0 & ~(C - 1);
            const int lo = 1 + kb * A, n_win = (k0 - kb + 1) * A;
            __syncthreads();
            for (int i = lane; i < n_win; i += 64) win[i] = NA[lo + i].L;
            const int ek = lane <= k0 - kb ? exp_map[kb + lane] : 0;
            __syncthreads();
            for (int k = k0; k >= kb; --k) {
                const int g = (k - kb) * A;
                const double mine = lane < A ? win[g + lane] : ninf;
                const double m = A <= 16 ? row0_max(mine) : wave_max(mine);
                const int parent_k = __builtin_amdgcn_readlane(ek, k - kb);
                if (lane == 0) {
                    NA[parent_k].L = m;
                    if (parent_k >= lo) win[parent_k - lo] = m;
                }
                __builtin_amdgcn_wave_barrier();
            }
            k0 = kb - 1;
        }
        __syncthreads();
        // ---- get_plan with DeterministicNode.selection_rule: a level is ONE round trip (the children's lower bounds and
        // their slots of the bounds array together); the chosen child's slot says which expansion made its children
        Pcg64 gen;
        gen.load(q->rng + (long)root * 6);
        int len = 0;
        int kcur = k_done > 0 ? 0 : -1;
        while (kcur >= 0) {
            const int fc = 1 + kcur * A;
            const double l = lane < A ? NA[fc + lane].L : ninf;
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
            kcur = ((unsigned)shi == 0xFFF80000u) ? slo : -1; // expanded: its k; a leaf: the plan ends
        }
        if (lane == 0) {
            gen.store(q->rng + (long)root * 6);
            if (plans)
                for (int i = len; i < max_plan_len; ++i) plans[(long)root * max_plan_len + i] = -1;
            if (plan_len) plan_len[root] = len;
            if (q->root_lower) q->root_lower[root] = NA[0].L;
            if (q->root_upper) q->root_upper[root] = root_upper;
        }
    } else if (lane == 0) {
        if (plans)
            for (int i = 0; i < max_plan_len; ++i) plans[(long)root * max_plan_len + i] = -1;
        if (plan_len) plan_len[root] = 0;
    }
    int n_real = real_mine;
    for (int off = 32; off > 0; off >>= 1) n_real += __shfl_xor(n_real, off);
    if (lane == 0) {
        if (q->status) q->status[root] = status;
        if (q->env_steps) q->env_steps[root] = (int64_t)n_real;
        q->n_nodes_out[root] = n_nodes;
    }
    for (int k = k_done + lane; k < q->K; k += 64) exp_map[k] = -1;
#undef LU
}

// ---- any number of actions (|A| > 64): the plain form.  One root per wavefront, bounds / records / parent map in global
// memory, the |A| children of an expansion 64 at a time, the leaf argmax a scan of the whole bounds array (the reference's
// own max(leaves, key=U), deterministic.py:110), backups and plan descent as in the other kernels -- same results, none of
// their machinery: environments with that many actions have small K = budget // |A|.
__global__ __launch_bounds__(64) void opd_any_kernel(OpdArgs p)
{
    const int lane = threadIdx.x, root = blockIdx.x, A = p.A;
    const long base = (long)root * p.cap;
    OpdNode *NA = reinterpret_cast<OpdNode *>(p.L) + base;
    double *U = p.U + base, *RW = p.reward + base;
    int32_t *EXP = p.expanded + (long)root * (p.K > 0 ? p.K : 1);
    constexpr int32_t DONE_FLAG = 1 << 30;
    const uint32_t done_bit = p.done_on_next ? 2u : 1u;
    const double ninf = -INFINITY;
    if (lane == 0) {
        OpdNode n0;
        n0.L = 0.0; n0.state = p.root_state[root]; n0.depth = 0;
        NA[0] = n0;
        RW[0] = 0.0;
        U[0] = 0.0;
    }
    __syncthreads();
    int n_nodes = 1, real_mine = 0, status = MP_OK, k_done = 0;
    for (int k = 0; k < p.K; ++k) {
        // deterministic.py:110: first maximal upper bound among the leaves (an expanded node's slot holds -inf)
        double bu = ninf;
        int leaf = 0x7fffffff;
        for (int i = lane; i < n_nodes; i += 64) {
            const double u = U[i];
            if (u > bu) { bu = u; leaf = i; }
        }
        wave_argmax(bu, leaf);
        if (leaf == 0x7fffffff) { status = MP_ERR_ARG; break; } // (no leaf with a finite bound: cannot happen with >= 1 action per state)
        const OpdNode pn = NA[leaf];
        const int d = (pn.depth & (DONE_FLAG - 1)) + 1;
        const double g1d = p.g1[d], gdivd = p.gdiv[d], tdivd = p.tdiv[d];
        const int g = n_nodes;
        bool bad = false;
        for (int a = lane; a < A; a += 64) { // DeterministicNode.expand / update, deterministic.py:28-65
            const Rec rc = p.rec[(long)pn.state * A + a];
            const double r = rc.reward;
            const bool avail = (rc.flags & 4u) != 0;      // deterministic.py:32-35 (phantom slots: see opd_kernel)
            bad |= avail && (!(0.0 <= r) || !(r <= 1.0)); // deterministic.py:46-47
            const bool dn = (rc.flags & done_bit) != 0;
            double Lc = pn.L + g1d * r;
            double Uc = Lc + gdivd;
            if (dn) {
                const double nv = Lc + tdivd;
                Lc = nv; Uc = nv;
            }
            if (!avail) { Lc = ninf; Uc = ninf; }
            OpdNode cn;
            cn.L = Lc; cn.state = rc.next; cn.depth = d | (dn ? DONE_FLAG : 0);
            NA[g + a] = cn;
            RW[g + a] = r;
            U[g + a] = Uc;
            real_mine += avail ? 1 : 0;
        }
        if (lane == 0) { U[leaf] = ninf; EXP[k] = leaf; }
        n_nodes += A;
        k_done = k + 1;
        if (ballot64(bad) != 0ull) { status = MP_ERR_REWARD_RANGE; break; }
        __syncthreads();
    }
    __syncthreads();
    double root_upper = ninf;
    for (int i = lane; i < n_nodes; i += 64) {
        const double u = U[i];
        if (u > root_upper) root_upper = u;
    }
    root_upper = wave_max(root_upper);
    if (status == MP_OK) {
        // lower bounds: L[parent of expansion k] = max over its children, children before parents (reverse expansion order)
        for (int k = k_done - 1; k >= 0; --k) {
            const int fc = 1 + k * A;
            double m = ninf;
            for (int a = lane; a < A; a += 64) {
                const double l = NA[fc + a].L;
                m = l > m ? l : m;
            }
            m = wave_max(m);
            if (lane == 0) NA[EXP[k]].L = m;
            __syncthreads();
        }
        // get_plan (abstract.py:143-156) with DeterministicNode.selection_rule (deterministic.py:21-26)
        Pcg64 gen;
        gen.load(p.rng + (long)root * 6);
        int len = 0, node = 0;
        for (;;) {
            int kcur = -1; // the expansion that made `node`'s children, if any
            for (int k0 = 0; k0 < k_done && kcur < 0; k0 += 64) {
                const unsigned long long hit = ballot64(k0 + lane < k_done && EXP[k0 + lane] == node);
                if (hit) kcur = k0 + __ffsll((long long)hit) - 1;
            }
            if (kcur < 0) break;
            const int fc = 1 + kcur * A;
            double m = ninf;
            for (int a = lane; a < A; a += 64) {
                const double l = NA[fc + a].L;
                m = l > m ? l : m;
            }
            m = wave_max(m);
            int nt = 0;
            for (int a0 = 0; a0 < A; a0 += 64) nt += __popcll(ballot64(a0 + lane < A && NA[fc + a0 + lane].L == m));
            int pick = (int)gen.below((uint32_t)nt), act = 0;
            for (int a0 = 0; a0 < A; a0 += 64) {
                unsigned long long t = ballot64(a0 + lane < A && NA[fc + a0 + lane].L == m);
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
            if (p.root_lower) p.root_lower[root] = NA[0].L;
            if (p.root_upper) p.root_upper[root] = root_upper;
        }
    } else if (lane == 0) {
        if (p.plans)
            for (int i = 0; i < p.max_plan_len; ++i) p.plans[(long)root * p.max_plan_len + i] = -1;
        if (p.plan_len) p.plan_len[root] = 0;
    }
    int n_real = real_mine;
    for (int off = 32; off > 0; off >>= 1) n_real += __shfl_xor(n_real, off);
    if (lane == 0) {
        if (p.status) p.status[root] = status;
        if (p.env_steps) p.env_steps[root] = (int64_t)n_real;
        p.n_nodes_out[root] = n_nodes;
    }
    for (int k = k_done + lane; k < p.K; k += 64) EXP[k] = -1;
}

} // namespace mp

using namespace mp;

extern "C" {

int mp_opd_plan(mp_ctx *ctx, mp_model *model, int32_t n_roots, const int32_t *root_state, int32_t budget,
                double gamma, double terminal_reward, uint64_t *rng_state, int32_t max_plan_len, int32_t *plans,
                int32_t *plan_len, double *root_lower, double *root_upper, int64_t *env_steps, int32_t *status,
                int32_t mem);

int mp_opd_plan_models(mp_ctx *ctx, mp_model *model, int32_t n_roots, const int32_t *model_index, const int32_t *root_state,
                       int32_t budget, double gamma, double terminal_reward, uint64_t *rng_state, int32_t max_plan_len,
                       int32_t *plans, int32_t *plan_len, double *root_lower, double *root_upper, int64_t *env_steps,
                       int32_t *status, int32_t mem)
{
    if (!mem_valid(mem)) return fail(MP_ERR_ARG, "mp_opd_plan_models: unknown mem flags %d", mem);
    std::vector<int32_t> tmp;
    const int32_t *global = nullptr;
    MP_TRY(globalize_roots_arg(ctx, model, n_roots, model_index, root_state, mem, tmp, &global));
    return mp_opd_plan(ctx, model, n_roots, global, budget, gamma, terminal_reward, rng_state, max_plan_len, plans, plan_len,
                       root_lower, root_upper, env_steps, status, mem);
}

int mp_opd_plan(mp_ctx *ctx, mp_model *model, int32_t n_roots, const int32_t *root_state, int32_t budget,
                double gamma, double terminal_reward, uint64_t *rng_state, int32_t max_plan_len, int32_t *plans,
                int32_t *plan_len, double *root_lower, double *root_upper, int64_t *env_steps, int32_t *status,
                int32_t mem)
{
    if (!ctx || !model || !root_state || !rng_state) return fail(MP_ERR_ARG, "mp_opd_plan: NULL argument");
    if (!mem_valid(mem)) return fail(MP_ERR_ARG, "mp_opd_plan: unknown mem flags %d", mem);
    const int rmem = mem_rng(mem); // MP_MEM_RNG_DEVICE: the generator records are device-resident (mp_rng) also with host arrays
    mem = mem_arrays(mem);
    if (model->mode != MP_MODE_DETERMINISTIC)
        return fail(MP_ERR_MODE, "mp_opd_plan: model mode %d is not a deterministic table", model->mode);
    const int A = model->A;
    const bool any_a = A > 64; // more actions than lanes: the plain kernel (opd_any_kernel)
    if (n_roots < 1 || budget < 0 || max_plan_len < 0) return fail(MP_ERR_ARG, "mp_opd_plan: bad sizes");
    const int K = budget / A; // deterministic.py:118
    if (K > 0 && !(gamma != 1.0))
        return fail(MP_ERR_ARG, "mp_opd_plan: gamma = 1 (the reference divides by 1 - gamma, deterministic.py:53: ZeroDivisionError)");
    const long cap = 1 + (long)K * A;
    const int T = (int)((cap + 63) / 64) | 1;
    const size_t lds_full = (size_t)64 * T * sizeof(double) + (size_t)(K > 0 ? K : 1) * sizeof(int32_t);
        const size_t lds_bounds = (size_t)64 * T * sizeof(double); // bounds only, parent map in HBM (EXPG)
    // high-occupancy variant: LDS only holds the window of the closing lower-bound pass, `chunk` expansions x |A| doubles
    int chunk = 64;
    while (chunk > 1 && (size_t)chunk * A * sizeof(double) > 4096) chunk >>= 1;
    const size_t lds_win = (size_t)chunk * A * sizeof(double);
    // variant: LDS-resident bounds while every root of the batch fits on the chip that way, else high occupancy
    const char *force = getenv("MP_OPD_MODEL"); // "lds" / "global": test hook
    const long cus = ctx->prop.multiProcessorCount;
    const long lds_roots = cus * (long)((kLdsBytes - 1024) / (lds_full ? lds_full : 1));
    const long expg_roots = cus * (long)((kLdsBytes - 1024) / (lds_bounds ? lds_bounds : 1));
    bool glb = lds_bounds > kLdsBytes - 1024 || n_roots > expg_roots;
    if (force && force[0] == 'g') glb = true;
    if (force && force[0] == 'l' && lds_bounds <= kLdsBytes - 1024) glb = false;
    // bounds in LDS: keep the parent map there too while that costs no residency
    bool expg = !glb && (lds_full > kLdsBytes - 1024 || n_roots > lds_roots);
    if (force && !glb && force[1] == 'd' && force[2] == 's' && force[3] == 'x') expg = true; // "ldsx": test hook
    const size_t lds = glb ? lds_win : (expg ? lds_bounds : lds_full);
    MP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    // gamma-power tables, host libm (bit-equal to Python's float **); depth <= K + 1
    const int D = K + 2;
    std::vector<double> tab((size_t)3 * D);
    for (int d = 0; d < D; ++d) {
        tab[d] = d >= 1 ? pow(gamma, (double)(d - 1)) : 0.0;
        tab[D + d] = pow(gamma, (double)d) / (1 - gamma);
        tab[2 * D + d] = terminal_reward * pow(gamma, (double)d) / (1 - gamma);
    }
    double *d_tab = nullptr;
    MP_TRY(upload_tables(ctx, 2, tab, &d_tab));

    OpdArgs a;
    // high-occupancy variant, sibling layout (default; MP_OPD_WIDE=cls: the residue-class layout): groups of |A| slots,
    // ceil((K + 1) / 64) groups per row, one cache line of padding so that rows do not all start in the same channels
    int lgP = 0;
    while ((1 << lgP) < A) ++lgP;
    const int groups = (K + 1 + 63) / 64;
    const int Tsib = groups * A + 16;
    const char *wide_env = getenv("MP_OPD_WIDE");
    const bool sib = !(wide_env && wide_env[0] == 'c');
    a.n_roots = n_roots; a.S = model->S; a.A = A; a.K = K; a.cap = (int)cap; a.T = T; a.chunk = chunk; a.Tsib = Tsib; a.lgP = lgP;
    { const char *cl = getenv("MP_OPD_CLOSING"); a.closing_chain = cl && cl[0] == 'c'; }
    a.done_on_next = model->done_on_next; a.max_plan_len = max_plan_len;
    a.rec = model->rec;
    a.g1 = d_tab; a.gdiv = d_tab + D; a.tdiv = d_tab + 2 * D;
    const size_t nn = (size_t)n_roots * cap;
    MP_TRY(ws_get(ctx, WS_TREE0, 2 * nn, &a.L)); // OpdNode records, 16 B each
    MP_TRY(ws_get(ctx, WS_TREE1, nn, &a.U));
    MP_TRY(ws_get(ctx, WS_TREE2, nn, &a.reward));
    a.leaf_global = nullptr;
    if (glb && !any_a) MP_TRY(ws_get(ctx, WS_TREE3, (size_t)n_roots * 64 * (sib ? Tsib : T), &a.leaf_global));
    MP_TRY(ws_get(ctx, WS_TREE7, (size_t)n_roots * (K > 0 ? K : 1) + n_roots, &a.expanded));
    a.n_nodes_out = a.expanded + (size_t)n_roots * (K > 0 ? K : 1);
    ctx->tree.kind = 2; ctx->tree.n_roots = n_roots; ctx->tree.A = A; ctx->tree.cap = (int)cap; ctx->tree.K = K;

    int32_t *d_rs = nullptr;
    MP_TRY(stage_in(ctx, WS_IO0, root_state, (size_t)n_roots, mem, &d_rs));
    a.root_state = d_rs;
    MP_TRY(stage_in(ctx, WS_IO2, (const uint64_t *)rng_state, (size_t)n_roots * 6, rmem, &a.rng));
    MP_TRY(stage_out_alloc(ctx, WS_IO3, plans, (size_t)n_roots * max_plan_len, mem, &a.plans));
    MP_TRY(stage_out_alloc(ctx, WS_IO4, plan_len, (size_t)n_roots, mem, &a.plan_len));
    MP_TRY(stage_out_alloc(ctx, WS_IO5, root_lower, (size_t)n_roots, mem, &a.root_lower));
    MP_TRY(stage_out_alloc(ctx, WS_IO6, root_upper, (size_t)n_roots, mem, &a.root_upper));
    MP_TRY(stage_out_alloc(ctx, WS_IO7, status, (size_t)n_roots, mem, &a.status));
    MP_TRY(stage_out_alloc(ctx, WS_IO8, env_steps, (size_t)n_roots, mem, &a.env_steps));

    // every finite bound is >= +0.0 (rewards are range-checked, gamma in [0, 1), terminal reward >= 0): the cheaper cross-lane
    // maxima (wave.hpp); MP_OPD_LOOP=0: the general ones always -- test hook, shared with mp_ropd_plan
    const char *nn_env = getenv("MP_OPD_LOOP");
    const bool nonneg = gamma >= 0 && gamma < 1 && terminal_reward >= 0 && !(nn_env && nn_env[0] == '0');
    const void *kfn = expg ? (nonneg ? (const void *)opd_kernel<true, true> : (const void *)opd_kernel<true, false>)
                           : (nonneg ? (const void *)opd_kernel<false, true> : (const void *)opd_kernel<false, false>);
    if (!any_a && lds > 64 * 1024) MP_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MP_TRY(kernels_begin(ctx));
    if (any_a) hipLaunchKernelGGL(opd_any_kernel, dim3((unsigned)n_roots), dim3(64), 0, st, a);
    else if (glb) {
        const bool small = sib ? groups * A <= 128 : a.T <= 128; // at most two slots per lane in a re-scan
        void (*kw)(OpdArgs) =
            sib ? (nonneg ? (small ? opd_wide_kernel<true, true, true> : opd_wide_kernel<true, false, true>)
                          : (small ? opd_wide_kernel<false, true, true> : opd_wide_kernel<false, false, true>))
                : (nonneg ? (small ? opd_wide_kernel<true, true, false> : opd_wide_kernel<true, false, false>)
                          : (small ? opd_wide_kernel<false, true, false> : opd_wide_kernel<false, false, false>));
        hipLaunchKernelGGL(kw, dim3((unsigned)n_roots), dim3(64), lds, st, a);
    }
    else if (expg && nonneg) hipLaunchKernelGGL((opd_kernel<true, true>), dim3((unsigned)n_roots), dim3(64), lds, st, a);
    else if (expg) hipLaunchKernelGGL((opd_kernel<true, false>), dim3((unsigned)n_roots), dim3(64), lds, st, a);
    else if (nonneg) hipLaunchKernelGGL((opd_kernel<false, true>), dim3((unsigned)n_roots), dim3(64), lds, st, a);
    else hipLaunchKernelGGL((opd_kernel<false, false>), dim3((unsigned)n_roots), dim3(64), lds, st, a);
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

int mp_opd_tree_export(mp_ctx *ctx, int32_t root, int32_t cap, int32_t *n_nodes, int32_t *parent, int32_t *action,
                       int32_t *state, int32_t *depth, double *reward, double *lower, double *upper, uint8_t *done,
                       int64_t *count, int32_t *first_child, int32_t *n_children)
{
    if (!ctx) return fail(MP_ERR_ARG, "ctx is NULL");
    if (ctx->tree.kind != 2) return fail(MP_ERR_ARG, "mp_opd_tree_export: no OPD tree on this ctx");
    if (root < 0 || root >= ctx->tree.n_roots) return fail(MP_ERR_ARG, "mp_opd_tree_export: root %d out of range", root);
    const int tcap = ctx->tree.cap, A = ctx->tree.A, K = ctx->tree.K, NR = ctx->tree.n_roots;
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
    std::vector<int32_t> fc((size_t)n, -1), exp((size_t)(K > 0 ? K : 1));
    MP_HIP(hipMemcpy(exp.data(), d_exp + (size_t)root * (K > 0 ? K : 1), (size_t)(K > 0 ? K : 1) * sizeof(int32_t),
                     hipMemcpyDeviceToHost));
    // the k-th expansion created node slots 1 + kA .. 1 + kA + A - 1 under exp[k]
    for (int k = 0; k < K && 1 + (k + 1) * A <= n; ++k)
        if (exp[k] >= 0 && exp[k] < n) fc[exp[k]] = 1 + k * A;
    std::vector<double> up((size_t)n), rw((size_t)n);
    std::vector<OpdNode> na((size_t)n);
    MP_TRY(pull(up.data(), WS_TREE1, sizeof(double)));
    MP_TRY(pull(rw.data(), WS_TREE2, sizeof(double)));
    MP_TRY(pull(na.data(), WS_TREE0, sizeof(OpdNode)));
    std::vector<int32_t> par((size_t)n);
    par[0] = -1;
    for (int i = 1; i < n; ++i) par[i] = exp[(i - 1) / A];
    // the kernel stores leaf upper bounds only (-inf marks an expanded node): U[n] = max over children,
    // bottom-up in reverse creation order (children have larger ids than their parent)
    for (int i = n - 1; i >= 0; --i)
        if (fc[i] >= 0) {
            double m = up[fc[i]];
            for (int a = 1; a < A; ++a)
                if (up[fc[i] + a] > m) m = up[fc[i] + a];
            up[i] = m;
        }
    // slots of unavailable actions (deterministic.py:32-35) are phantoms with L = -inf: not nodes of the tree
    std::vector<int32_t> id((size_t)n, -1);
    int kept = 0;
    for (int i = 0; i < n; ++i)
        if (!(na[i].L == -INFINITY)) id[i] = kept++;
    if (kept > cap) return fail(MP_ERR_ARG, "mp_opd_tree_export: capacity %d < %d nodes", cap, kept);
    // deterministic.py:62-63: every node on the root..child sequence gets +1 per created child;
    // count = 1 (initial) + size of own subtree for non-root nodes, root: 1 + #descendants
    std::vector<int64_t> sz((size_t)n, 0);
    for (int i = n - 1; i >= 0; --i) {
        if (id[i] < 0) continue;
        sz[i] += 1;
        if (i > 0) sz[par[i]] += sz[i];
    }
    for (int i = 0; i < n; ++i) {
        if (id[i] < 0) continue;
        const int o = id[i];
        if (lower) lower[o] = na[i].L;
        if (state) state[o] = na[i].state;
        if (depth) depth[o] = na[i].depth & ((1 << 30) - 1);
        if (done) done[o] = (uint8_t)((na[i].depth >> 30) & 1);
        if (upper) upper[o] = up[i];
        if (reward) reward[o] = rw[i];
        if (parent) parent[o] = i == 0 ? -1 : id[par[i]];
        if (action) action[o] = i == 0 ? -1 : (i - 1) % A;
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
    }
    if (n_nodes) *n_nodes = kept;
    return MP_OK;
}

} // extern "C"
