// uct.hip -- MCTS/UCT planning (tree_search/mcts.py) for thousands of independent roots.
//
// Mapping: ONE ROOT PER LANE.  A root's plan() is inherently sequential (every episode reads
// the tree the previous one updated; every env step depends on the previous state), so the
// parallel axis is roots.  A cloned environment (reference: safe_deepcopy_env,
// common/factory.py:119-134) is an (int32 state, int32 steps) register pair.  A batch takes as
// long as its slowest lane's dependency chain, so the kernel is built to shorten that chain:
//
//   model   default: one 16-byte record {next, flags, reward} per (s,a), a single dwordx4 gather per
//           env step (any S).  MP_UCT_MODEL=lds (S < 32768 and S*A*2 B fits): the transition table is
//           staged once per workgroup into LDS as uint16 {bit15 = terminal[next], next state} and the
//           step's reward is fetched off the chain, added one step later in the reference's order.
//           Measured slower (0.376 vs 0.338 ms at 4096 roots): the chain is instruction-bound.
//           ENV_TABLE_LDSR (round 4, the default wherever the model fits): the WHOLE model lives in LDS -- uint16
//           transitions, a uint8 index per (s,a) into the table of the model's distinct rewards (<= 256 of them) and that
//           table -- 3 B per (s,a): the headline's S*A = 50 000 takes 150 of the CU's 160 KB.  An env step then makes
//           no global-memory request at all; the path stack moves from LDS to registers (depths 1..5) with a global
//           spill beyond.  One workgroup of up to 16 waves per CU shares the tables.
//           ENV_CARTPOLE: closed-form dynamics, state in registers.
//   rng     numpy PCG64 stepped per lane (pcg64.hpp, 128-bit multiply on 32-bit limbs); the draw for
//           step h+1 is computed on a second generator copy while step h's lookup is in flight and
//           kept only if the rollout continues, so the stream stays bit-identical to the reference's.
//           Rollout actions come from integer thresholds ceil(cdf * 2^53) <= (next64 >> 11).
//   tables  gamma**h, the thresholds, 1/n and temperature*|A|*prior[a]/n for every visit count n of a
//           fresh tree are computed on the host with the reference's own operations (libm pow, IEEE
//           divide) and read from LDS; larger counts (kept trees) use the device's IEEE division.
//   tree    Node[n_roots][cap] root-major 16-byte records {value f64, count i32, first_child i32}
//           in HBM; the A children of a node are contiguous.  cap = 1 + episodes*A (+ the kept tree).
//           The root and its children (ids 0..A) live in registers for the whole plan, and the path
//           nodes of depths 2..5 keep the statistics their selection read until the backup: the
//           saturated kernel is bound by L2/HBM requests, and these remove a third of them.
//   path    per-lane stack of visited node ids in LDS ([depth][lane], conflict-free): the backup
//           walks known addresses, not parent pointers.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include <type_traits>

#include "common.hpp"
#include "libm_sincos.hpp"
#include "pcg64.hpp"
#include "wave.hpp"

namespace mp {

struct alignas(16) UctNode {
    double value;
    int32_t count;
    int32_t first_child; // -1 = not expanded
};
static_assert(sizeof(UctNode) == 16, "UctNode must be one dwordx4");

// One root's tree inside the batch buffer.  LAY selects the layout (A/B: profiles/r02_uct_tree_layout.md):
//   0 root-major       Node[n_roots][cap]: a root's nodes contiguous -- the |A| children of a node share one or two
//                      cache lines, but neighbouring lanes are a whole tree apart: nothing a wave does coalesces.
//   1 interleaved      Node[n_roots / 64][cap][64]: the same node id of a wavefront's 64 roots contiguous -- accesses
//                      that hit the same id in every lane (expansions: ids advance in lock-step) fill whole lines, but a
//                      node's |A| children now lie in |A| different lines.
//   2 group-interleaved Node[n_roots / 64][groups][64][|A|]: node ids come in groups of |A| siblings (1 + g|A| .. ;
//                      the root fills the last slot of group 0); a lane's sibling group stays contiguous (one or two
//                      lines per scored level, as root-major) AND the same group of the 64 lanes is contiguous
//                      (expansions write 64 x |A| x 16 B = one run).  Needs |A| at compile time.
// Hot sites address a child through its sibling group (child / handle: one add and a shift); operator[] maps any id.
template <int LAY, int AT>
struct TreeRef {
    UctNode *base;
    int A; // runtime |A| (generic ids only)
    __device__ __forceinline__ static unsigned idx01(int n) { return LAY == 1 ? (unsigned)n << 6 : (unsigned)n; }
    // node fc + a where fc is the first child of some node (fc = 1 mod |A|)
    __device__ __forceinline__ int handle(int fc, int a) const { return LAY == 2 ? ((fc - 1 + AT) << 6) + a : fc + a; }
    __device__ __forceinline__ int root_handle() const { return LAY == 2 ? AT - 1 : 0; }
    __device__ __forceinline__ UctNode &at(int h) const { return base[LAY == 2 ? (unsigned)h : idx01(h)]; }
    __device__ __forceinline__ UctNode &child(int fc, int a) const { return at(handle(fc, a)); }
    // sibling a of a group whose first node is *first (= &child(fc, 0)): one address computation per group, the siblings
    // at constant offsets (the compiler cannot derive that from 32-bit handles: it re-materialises a 64-bit address each)
    __device__ __forceinline__ static UctNode &sibling(UctNode *first, int a) { return first[LAY == 1 ? a << 6 : a]; }
    __device__ __forceinline__ UctNode &operator[](int n) const
    {
        if (LAY != 2) return base[idx01(n)];
        const int x = n + A - 1, q = x / A;          // group q (root: group 0, last slot), slot x - q|A|
        return base[(unsigned)(q * A * 64 + (x - q * A))];
    }
};
template <int LAY, int AT>
__device__ __forceinline__ TreeRef<LAY, AT> tree_of(UctNode *trees, int r, int cap, int A)
{
    TreeRef<LAY, AT> t;
    t.A = A;
    if (LAY == 2) t.base = trees + ((long)(r >> 6) * (cap + A) * 64 + (long)(r & 63) * A);
    else if (LAY == 1) t.base = trees + ((long)(r >> 6) * cap * 64 + (r & 63));
    else t.base = trees + (long)r * cap;
    return t;
}
// nodes to allocate per root for a layout
static inline size_t tree_stride_alloc(int lay, long cap, int A) { return (size_t)(lay == 2 ? cap + A : cap); }

struct UctArgs {
    int n_roots, S, A, episodes, horizon, cap;
    int table_n; // counts with entries in the quotient tables of `tab` (TE below): min(episodes, what 16 KB of LDS hold)
    int tree_il; // tree layout (TreeRef): 0 root-major, 1 interleaved, 2 group-interleaved
    int done_on_next, max_steps, max_plan_len;
    int lanes; // roots per wavefront (64 = dense; fewer spreads a small batch over more SIMDs)
    int rep_shift; // CartPole: a root is replicated over 2^rep_shift lanes (see the launch code: wavefronts of few ACTIVE lanes run slow)
    int waves; // wavefronts per workgroup
    int Sb;    // batch models: states per MDP (root r plans on MDP root_state[r] / Sb); = S for any other model
    const Rec *rec;
    const uint16_t *t16; // compact transitions (LDS variants), [S*A]
    const uint8_t *r8;   // LDS-resident variant: index of reward[s, a] in rdict, [S*A]
    const double *rdict; //                       the model's distinct reward values, [n_rdict] (<= 256)
    int n_rdict;
    const uint32_t *jump; // four-lanes-per-root variant: limbs of A^n and G_n, n = 0..H (PCG64 jump-ahead), [H + 1][8]
    int32_t *path_spill; // LDS-resident variant: path handles of depths the registers do not hold, [H + 1][spill_stride]
    long spill_stride;
    const int32_t *root_state, *root_steps;
    const double *root_x; // CartPole roots: [n_roots][4] = x, x_dot, theta, theta_dot
    mp_cartpole_params cp;
    int cp_sincos;       // how CartPole's sin / cos are evaluated (libm_sincos.hpp: 0 device, 1 / 2 the host libm's forms)
    int cp_fastdiv;      // 1: the parameters qualify for div_by_const (host check in the launch code)
    double cp_inv_tm;    // RN(1 / total_mass)
    const double *tab; // gpow[H+1] | thr[A] (uint64 bits) | tp[A] | rcp[TE+1] | tpdiv[A][TE+2]
    uint64_t thr_arg[8]; // the same thresholds by value (|A| <= 8), SHIFTED UP by 11 bits (compared with the raw 64-bit draw; 2^53 -> ~0): kernel arguments live in SGPRs
    int thr_valid;       // how many of the first |A| - 1 thresholds are below 2^53 (the others can never be reached)
    // per-state policies (mp_policy): nullptr / 0 for the state-independent ones
    const double *pol_prior; // [S][pol_stride]
    const uint64_t *pol_thr; // [S][pol_stride]
    const uint4 *pol_frec;   // [S*A][1 + ceil((A-1)/4)]
    const uint4 *pol_frec_roll; // the same records laid out by ROLLOUT slot (= pol_frec unless the rollout policy lists the actions in another order)
    int pol_stride;
    int pol_shift;           // fused thresholds = min(thr >> pol_shift, pol_sat); 21 = the top 32 of 53 bits
    uint32_t pol_sat;        // 2^32 - 1, or 1023 for the packed 16-byte records
    int pol_packed;          // 1: 16-byte records {next:20 | th0:10, flags:2 | th1:10 | th2:10 | th3:10, reward f64}
    double TA;               // temperature * |A|  (mcts.py:286, left to right)
    double temperature;      // listed policies: the factor is temperature * len(children) of the node being scored
    uint64_t *rng;
    UctNode *tree;
    const int32_t *n_nodes_in; // kept (re-rooted) tree sizes, nullptr = every root starts fresh
    int32_t *n_nodes_out;
    int32_t *plans, *plan_len;
    double *root_value, *root_child_value;
    int64_t *root_child_count, *env_steps;
};


// a / b for a CONSTANT b whose correctly rounded reciprocal y = RN(1 / b) the host supplies, as five multiply-adds: q0 = RN(a y) is
// within two ulps of a / b, q1 = RN(q0 + r0 y) with the exact residual r0 = a - b q0 lies within 2^-100 of a / b before its rounding,
// hence is a faithful quotient, and q2 = RN(q1 + r1 y) is then the CORRECTLY ROUNDED a / b (Markstein 1990; Muller et al., Handbook
// of Floating-Point Arithmetic, ch. 4.7: holds unless b's significand is all ones) -- the same bits as the IEEE division, which
// the hardware does in thirteen instructions (v_div_scale x 2, v_rcp, four refinements, v_div_fmas, v_div_fixup).  Exact while
// the residuals are: a = 0, or 2^-960 <= |a| <= 2^1000 (the CartPole rollout proves that range from a bound on the pole's
// velocity it tracks; see there); b normal.
__device__ __forceinline__ double div_by_const(double a, double b, double y)
{
    const double q0 = a * y;
    const double r0 = __fma_rn(-q0, b, a);
    const double q1 = __fma_rn(r0, y, q0);
    const double r1 = __fma_rn(-q1, b, a);
    return __fma_rn(r1, y, q1);
}
// lane I of every DPP row (16 lanes), to the row's lanes (row_newbcast:I)
template <int I>
__device__ __forceinline__ uint32_t row_bcast(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x150 + I, 0xf, 0xf, false);
}
// lane Q of every quad, to the quad's four lanes (DPP quad_perm Q,Q,Q,Q)
template <int Q>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, Q * 0x55, 0xf, 0xf, true);
}

// rl_agents_amd/envs/cartpole.py step(), operation for operation in IEEE double (no contraction).
// sin / cos of the pole angle: the HOST libm's algorithm restated (libm_sincos.hpp; `sincos_mode` = the form that
// reproduces this host's sin / cos, found at model-load time) -- bit-exact plans by construction since round 5; mode 0 = the
// device math library (last bit may differ from glibc's: the tolerance of rounds 1-4).
__device__ __forceinline__ bool cartpole_step(const mp_cartpole_params &c, double (&x4)[4], int act, int sincos_mode, const double *sctab)
{
    const double total_mass = c.masspole + c.masscart, polemass_length = c.masspole * c.length;
    double x = x4[0], x_dot = x4[1], theta = x4[2], theta_dot = x4[3];
    const double force = act == 1 ? c.force_mag : -c.force_mag;
    double sintheta, costheta;
    libm_sincos(sincos_mode, theta, &sintheta, &costheta, sctab);
    const double temp = (force + polemass_length * (theta_dot * theta_dot) * sintheta) / total_mass;
    const double thetaacc = (c.gravity * sintheta - costheta * temp) /
                            (c.length * (4.0 / 3.0 - c.masspole * (costheta * costheta) / total_mass));
    const double xacc = temp - polemass_length * thetaacc * costheta / total_mass;
    x = x + c.tau * x_dot;
    x_dot = x_dot + c.tau * xacc;
    theta = theta + c.tau * theta_dot;
    theta_dot = theta_dot + c.tau * thetaacc;
    x4[0] = x; x4[1] = x_dot; x4[2] = theta; x4[3] = theta_dot;
    return x < -c.x_threshold || x > c.x_threshold || theta < -c.theta_threshold || theta > c.theta_threshold;
}

enum { ENV_TABLE = 0, ENV_TABLE_LDS = 1, ENV_CARTPOLE = 2, ENV_TABLE_LDSR = 3, ENV_TABLE_SPILL = 4 };
// ENV_TABLE_SPILL: ENV_TABLE with the path stack in registers + a global spill array instead of LDS -- for horizons whose
// [H + 1][64] stack would not fit (the reference has no horizon limit: tree_search/mcts.py:116-118)

// AT > 0: |A| known at compile time (children scored from registers in one pass);
// AT == 0: any |A| (three passes over the children).  ENV: where an env step comes from.
#ifndef MP_UCT_MIN_WAVES
#define MP_UCT_MIN_WAVES 1
#endif
// SP: the prior and the rollout distribution are rows of per-state tables (mcts_with_prior.py:47-62).  The env is
// deterministic, so the state a node is reached in is a function of its action sequence and the prior stored in a
// child at expansion (mcts.py:237-246) is prior[state of the parent][action]: it is looked up with the state the
// descent is in, not stored.  Rollouts read one fused 32-byte (|A| <= 5) or 48-byte record per step -- the (s,a)
// record followed by the sampling thresholds of the state it leads to -- so the per-step dependency chain stays one
// gather long.  The fused thresholds are the top 32 of the 53 bits: they decide the draw unless one of them equals
// the draw's top 32 bits (probability ~|A| * 2^-32 per step), in which case the exact 53-bit row is fetched.
// MK (with SP): the prior policy lists only a subset of the actions per state (restricted action sets, mcts.py:59-97).  An
// expansion still appends |A| node slots so that ids stay lock-step across lanes, but the slots of unlisted actions are
// PHANTOMS (count = -1): never scored, never visited, dropped by the tree export.  The listed-action masks of the state
// reached / acted from ride in bits 8-15 / 16-23 of the fused records' flags word, so the descent always knows the mask
// of the state it is in without another gather; len(children) in the exploration term is the mask's population count.
// QD (with ENV_TABLE_LDSR; round 5): FOUR LANES PER ROOT, for batches too small to fill the chip (SURVEY 8(d)'s own sizes: 4096
// roots, one root).  A lone wave per SIMD runs at the latency of its dependency chain, and in a rollout step that chain is
// numpy's PCG64 (27 of the step's 52 vector instructions) + the model lookup.  With state-independent policies the ACTION
// SEQUENCE of a rollout does not depend on the states it visits: the four lanes of a root draw the rollout's H actions four
// at a time -- lane j holds the generator j + 1 steps ahead and jumps by four per round (state <- A^4 state + inc G_4: one
// 128-bit multiply, as a single step) -- and the root's first lane (its OWNER: tree phases are its alone) walks the model in
// LDS through each round's four actions while the next round's draws are computed beside it: per step one LDS lookup on the
// state chain and no generator arithmetic.  The walk consumes n draws; the generator then jumps by exactly n (limbs of A^n,
// G_n from a table), so the stream stays the reference's.  Same trees, plans and generator states as every other variant.
template <int AT, int ENV, bool SP = false, bool MK = false, int IL = 0, bool QD = false>
// Per-state-policy kernels with |A| <= 5 are held to the registers of 4 waves per SIMD (they would take 134-140 VGPRs = 3
// waves; TA 52 %, VALU 45 %, L1 28 % busy: latency-bound -- 2.14 -> 1.91 ms at 262 144 roots with the fourth wave).
__global__ __launch_bounds__((ENV == ENV_TABLE_LDS || ENV == ENV_TABLE_LDSR) ? 1024 : (ENV == ENV_CARTPOLE ? 256 : 64),
                             (ENV == ENV_TABLE_LDS || ENV == ENV_TABLE_LDSR) ? 1 : (SP && AT > 0 && AT <= 5 && MP_UCT_MIN_WAVES < 4 ? 4 : MP_UCT_MIN_WAVES))
void uct_kernel(UctArgs p)
{
    static_assert(!SP || (AT > 0 && ENV == ENV_TABLE), "per-state policies: table env, |A| known at compile time");
    static_assert(!MK || SP, "listed policies are per-state policies");
    constexpr int NTH = AT > 1 ? AT - 1 : 1; // thresholds that can be reached (the last one is 2^53: never)
    constexpr int NQ = (NTH + 1) / 2;        // 16-byte chunks of a row of exact (uint64) thresholds
    constexpr int NQ32 = (NTH + 3) / 4;      // 16-byte chunks of the fused record's 32-bit thresholds
    constexpr bool LDSR = ENV == ENV_TABLE_LDSR;             // transitions AND rewards in LDS, path stack in registers
    constexpr bool LDSM = ENV == ENV_TABLE_LDS || LDSR;      // transitions in LDS (reward one step behind the state chain)
    static_assert(!LDSR || (AT > 0 && !SP), "the LDS-resident variant: |A| at compile time, state-independent policies");
    static_assert(!QD || LDSR, "four lanes per root: the LDS-resident model");
    constexpr bool PREG = LDSR || ENV == ENV_TABLE_SPILL;    // path stack in registers + global spill, not in LDS
    constexpr bool CART = ENV == ENV_CARTPOLE;
    // RAWU: the rollout compares the generator's raw 64-bit output with thresholds shifted up by 11 bits (thr <= out >> 11
    // <=> thr << 11 <= out) instead of shifting every draw down to its 53 random bits
    constexpr bool RAWU = AT > 0 && !SP;
    constexpr int USH = RAWU ? 0 : 11;
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int A = AT > 0 ? AT : p.A, H = p.horizon, E = p.episodes;
    const int nthreads = p.waves * 64;
    double *gpow = lds_d;                   // [H + 1]  gamma ** h
    // rollout policy as integer thresholds: u = k * 2^-53 with k = next64 >> 11, and
    // searchsorted(cdf, u, 'right') = #{a : cdf[a] <= u} = #{a : ceil(cdf[a] * 2^53) <= k}
    const uint64_t *thr = reinterpret_cast<const uint64_t *>(gpow + (H + 1)); // [A]
    double *tp = gpow + (H + 1) + A;        // [A]      temperature * |A| * prior[a]
    const int TE = p.table_n;               // counts the quotient tables cover
    double *rcp = tp + A;                   // [TE + 1]  1.0 / n
    double *tpdiv = rcp + (TE + 1);         // [A][TE+2] temperature * |A| * prior[a] / n
    const int ntab = (H + 1) + 2 * A + (TE + 1) + A * (TE + 2);
    // visit counts beyond the tables (trees kept across plans, step_strategy "subtree"; plans of more episodes than the
    // tables hold) take the IEEE division itself -- the same correctly rounded quotient the host put in the tables
    auto explore = [&](int a, int cnt1) { return cnt1 <= TE + 1 ? tpdiv[a * (TE + 2) + cnt1] : tp[a] / (double)cnt1; };
    auto inv = [&](int c) { return c <= TE ? rcp[c] : 1.0 / (double)c; };
    int32_t *path_all = reinterpret_cast<int32_t *>(lds_d + ntab); // [H + 1][waves * 64]  (not in the LDS-resident variant)
    const int ntab2 = (ntab + 1) & ~1;                             // (16-byte alignment of what follows)
    const double *rdict = lds_d + ntab2;                           // LDSR: [n_rdict] distinct rewards
    uint16_t *t16 = LDSR ? reinterpret_cast<uint16_t *>(lds_d + ntab2 + ((p.n_rdict + 1) & ~1))
                         : reinterpret_cast<uint16_t *>(path_all + (H + 1) * nthreads);
    uint8_t *r8 = reinterpret_cast<uint8_t *>(t16 + ((p.S * A + 7) & ~7)); // LDSR: [S*A]
    uint32_t *jump = reinterpret_cast<uint32_t *>(r8 + ((p.S * A + 15) & ~15)); // QD: [H + 5][8]
    if (QD)
        for (int i = tid; i < (H + 5) * 8; i += nthreads) jump[i] = p.jump[i];
    // CartPole: the sin / cos table of libm_sincos.hpp, in LDS behind the path stack (the host sizes the allocation)
    double *sctab_lds = reinterpret_cast<double *>(path_all + (((H + 1) * nthreads + 1) & ~1));
    if (CART)
        for (int i = tid; i < MP_SINCOS_ENTRIES; i += nthreads) sctab_lds[i] = kSincosTab[i];
    // CartPole with replicated roots: the generator's jump table behind the sin / cos table
    uint32_t *cjump = reinterpret_cast<uint32_t *>(sctab_lds + MP_SINCOS_ENTRIES); // [H + 5][8]
    if (CART && p.rep_shift >= 2)
        for (int i = tid; i < (H + 5) * 8; i += nthreads) cjump[i] = p.jump[i];
    const double *sctab = CART ? sctab_lds : nullptr;
    for (int i = tid; i < ntab; i += nthreads) lds_d[i] = p.tab[i];
    if (LDSR) {
        for (int i = tid; i < p.n_rdict; i += nthreads) lds_d[ntab2 + i] = p.rdict[i];
        const int n16 = (p.S * A + 15) >> 4; // (the device array is padded to whole 16-byte chunks)
        const uint4 *src = reinterpret_cast<const uint4 *>(p.r8);
        uint4 *dst = reinterpret_cast<uint4 *>(r8);
        for (int i = tid; i < n16; i += nthreads) dst[i] = src[i];
    }
    if (LDSM) {
        // S*A uint16 entries, staged with 16-byte loads where the tail allows
        const int n = p.S * A;
        const int n8 = n >> 3;
        const uint4 *src = reinterpret_cast<const uint4 *>(p.t16);
        uint4 *dst = reinterpret_cast<uint4 *>(t16);
        for (int i = tid; i < n8; i += nthreads) dst[i] = src[i];
        for (int i = (n8 << 3) + tid; i < n; i += nthreads) t16[i] = p.t16[i];
    }
    __syncthreads();
    int32_t *path = path_all + wave * 64; // slot d of this lane: path[d * nthreads + lane]
    const int slot = QD ? lane >> 2 : (CART ? lane >> p.rep_shift : lane);   // which root of the wave (QD: four lanes per root, the first one owns it)
    const bool owner = QD ? (lane & 3) == 0 : true;
    const int r = (blockIdx.x * p.waves + wave) * p.lanes + slot;
    if (slot >= p.lanes || r >= p.n_roots) return;   // (QD: whole quads leave together)
    static_assert(IL != 2 || AT > 0, "the group-interleaved layout needs |A| at compile time");
    const TreeRef<IL, AT> tree = tree_of<IL, AT>(p.tree, r, p.cap, A);
    const Rec *__restrict__ rec = p.rec;
    Pcg64 g;
    g.load(p.rng + (long)r * 6);
    uint64_t g4_lo = 0, g4_hi = 0;          // QD: inc * G_4, the additive term of a four-step jump of this root's generator
    if (QD || (CART && p.rep_shift >= 2)) g.inc_g4(g4_lo, g4_hi);
    uint64_t g16_lo = 0, g16_hi = 0;        // CartPole, sixteen replicas: inc * G_16
    if (CART && p.rep_shift >= 4) g.inc_g16(g16_lo, g16_hi);
    const int32_t s0 = CART ? 0 : p.root_state[r];
    const int32_t st0 = p.root_steps ? p.root_steps[r] : 0;
    const uint32_t done_bit = p.done_on_next ? 2u : 1u;
    // terminal flag of the root state itself ("source" rule): every record of a state carries it
    const bool root_term = CART ? false : (rec[(long)s0 * A].flags & 1u) != 0;
    double x0[4] = {0.0, 0.0, 0.0, 0.0};
    if (CART) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x0[i] = p.root_x[(long)r * 4 + i];
    }
    // The pipelined CartPole rollouts evaluate the RESTATED sin / cos without its range test (|x| < 0.855: libm_sincos.hpp).  A
    // live step's angle is within the env's threshold, so they are used only where the threshold and every root angle of the wave
    // are inside that range; anything else (a root handed over with the pole already down, a wide threshold) takes the generic
    // rollout, whose cartpole_step tests the range per call.
    const bool cart_flat = CART && p.cp_sincos != SINCOS_DEVICE && p.cp.theta_threshold < 0.8 && !any64(!(fabs(x0[2]) < 0.8));
    int n_nodes = p.n_nodes_in ? p.n_nodes_in[r] : 0; // > 0: tree kept by step_strategy "subtree"
    constexpr int AR = AT > 0 ? AT : 1;
    // RC: the root and its children (node ids 0..A: the root is always the first node to expand, and a re-rooted
    // tree is renumbered breadth-first) live in registers for the whole plan.  Every episode scores the root's
    // children and backs up the root and one of them: keeping them out of memory removes ~7 of the ~31 L2
    // requests an episode makes.  They are written back once, before the plan is read out.
    constexpr bool RC = AT > 0;
    double tv0 = 0.0, tv[AR];   // value of the root / of its children
    int tc0 = 0, tc[AR];        // count
    int tf0 = -1, tf[AR];       // first_child
#pragma unroll
    for (int a = 0; a < AR; ++a) { tv[a] = 0.0; tc[a] = 0; tf[a] = -1; }
    if (n_nodes < 1) {
        // mcts.py:129-130 reset(): fresh root
        UctNode n;
        n.value = 0.0; n.count = 0; n.first_child = -1;
        if (!RC) tree[0] = n;
        n_nodes = 1;
    } else if (RC) {
        const UctNode r0 = tree[0];
        tv0 = r0.value; tc0 = r0.count; tf0 = r0.first_child;
        if (tf0 == 1) {
#pragma unroll
            for (int a = 0; a < AR; ++a) {
                const UctNode ca = tree.child(1, a);
                tv[a] = ca.value; tc[a] = ca.count; tf[a] = ca.first_child;
            }
        }
    }
    double tp0[AR]; // SP: temperature * |A| * prior[s0][a], the root's children
    uint32_t mask0 = 0xffu; // MK: actions the prior policy lists in the root state
    if (MK) mask0 = (p.pol_frec[(size_t)((unsigned)s0 * A) * (1 + NQ32)].y >> 16) & 0xffu;
    const double TA0 = MK ? p.temperature * (double)__popc(mask0) : p.TA;
#pragma unroll
    for (int a = 0; a < AR; ++a) tp0[a] = SP ? TA0 * p.pol_prior[(long)s0 * p.pol_stride + a] : 0.0;
    int steps_taken = 0;
    // statistics of the path nodes at depths 2..5 as read by the selection of this episode (nobody else writes
    // them in between): the backup then needs no read for them (~16 % of the saturated kernel's time was those reads)
    constexpr int KEEP = (AT > 0 && AT <= 5) ? 4 : 2; // register budget: stay within 128 VGPRs (4 waves per SIMD)
    double kv[KEEP];
    int kc[KEEP];
#pragma unroll
    for (int i = 0; i < KEEP; ++i) { kv[i] = 0.0; kc[i] = 0; }
    // where the handles of the path nodes live: a per-lane stack in LDS ([depth][lane]) -- or, when the model fills the
    // LDS (LDSR), registers for depths 1 .. 1 + KEEP and a global spill array beyond (rarely reached: mean depth ~2)
    int ph1 = 0, ph[KEEP];
#pragma unroll
    for (int i = 0; i < KEEP; ++i) ph[i] = 0;
    // (macros, not lambdas: a closure over `ph` keeps a dead copy of the array in scratch memory)
#define PATH_PUT(d_, h_)                                                                        \
    do {                                                                                        \
        const int pd_ = (d_), phv_ = (h_);                                                      \
        if constexpr (!PREG) {                                                                  \
            path[pd_ * nthreads + lane] = phv_;                                                 \
        } else {                                                                                \
            if (pd_ == 1) ph1 = phv_;                                                           \
            _Pragma("unroll") for (int i_ = 0; i_ < KEEP; ++i_) ph[i_] = pd_ == 2 + i_ ? phv_ : ph[i_]; \
            if (pd_ >= 2 + KEEP) p.path_spill[(size_t)pd_ * p.spill_stride + r] = phv_;        \
        }                                                                                       \
    } while (0)
#define PATH_GET(out_, d_)                                                                      \
    do {                                                                                        \
        const int pd_ = (d_);                                                                   \
        if constexpr (!PREG) {                                                                  \
            out_ = path[pd_ * nthreads + lane];                                                 \
        } else {                                                                                \
            int hh_ = pd_ == 0 ? tree.root_handle() : ph1;                                      \
            _Pragma("unroll") for (int i_ = 0; i_ < KEEP; ++i_) hh_ = pd_ == 2 + i_ ? ph[i_] : hh_; \
            if (pd_ >= 2 + KEEP) hh_ = p.path_spill[(size_t)pd_ * p.spill_stride + r];          \
            out_ = hh_;                                                                         \
        }                                                                                       \
    } while (0)

#ifdef MP_PROFILE
    long long t_sel = 0, t_expd = 0, t_roll = 0, t_bak = 0, n_sel = 0, n_roll = 0;
    const long long t_all0 = clock64();
#define PROF_T(x) const long long x = clock64()
#else
#define PROF_T(x)
#endif
    for (int ep = 0; ep < E; ++ep) { // mcts.py:179-184
        PROF_T(c0);
        int32_t s = s0, st = st0;
        double x4[4] = {x0[0], x0[1], x0[2], x0[3]};
        int node = 0, depth = 0;
        bool terminal = false;
        bool cur_term = root_term; // terminal[s] of the state the next action is taken from
        double total = 0.0;
        uint32_t cur_mask = mask0; // MK: listed actions of the state the descent is in
        if (!RC) PATH_PUT(0, tree.root_handle());
        int hnode = tree.root_handle(); // where `node` lives (TreeRef handle)
        int fc = RC ? tf0 : tree[0].first_child;
        // ---- selection, mcts.py:143-149
        while (owner && depth < H && fc >= 0 && !terminal) {
            // MCTSNode.selection_strategy (mcts.py:275-286): value + temperature*|A|*prior/(count+1);
            // Node.random_argmax (abstract.py:296-311): exact-equality argmax set, uniform draw
            // among >= 2 ties (no draw for a single maximum)
            int act = 0, nfc = -1;
            if (AT > 0) {
                UctNode c[AR];
                if (fc == 1) { // the root's children: registers
#pragma unroll
                    for (int a = 0; a < AR; ++a) { c[a].value = tv[a]; c[a].count = tc[a]; c[a].first_child = tf[a]; }
                } else {
                    UctNode *grp = &tree.child(fc, 0);
#pragma unroll
                    for (int a = 0; a < AR; ++a) c[a] = tree.sibling(grp, a);
                }
                double sc[AR];
                if (SP) {
                    double tpl[AR];
                    if (depth == 0) {
#pragma unroll
                        for (int a = 0; a < AR; ++a) tpl[a] = tp0[a];
                    } else {
                        const double *pr = p.pol_prior + (long)s * p.pol_stride;
                        const double TAk = MK ? p.temperature * (double)__popc(cur_mask) : p.TA;
#pragma unroll
                        for (int a = 0; a < AR; ++a) tpl[a] = TAk * pr[a];
                    }
#pragma unroll
                    for (int a = 0; a < AR; ++a) {
                        sc[a] = c[a].value + tpl[a] / (double)(c[a].count + 1);
                        if (MK && c[a].count < 0) sc[a] = -INFINITY; // phantom slot: not a child
                    }
                } else {
#pragma unroll
                    for (int a = 0; a < AR; ++a) sc[a] = c[a].value + explore(a, c[a].count + 1);
                }
                double m = sc[0];
#pragma unroll
                for (int a = 1; a < AR; ++a) m = sc[a] > m ? sc[a] : m;
                int nt = 0;
#pragma unroll
                for (int a = 0; a < AR; ++a) nt += sc[a] == m ? 1 : 0;
                int pick = nt > 1 ? (int)g.below((uint32_t)nt) : 0;
                bool found = false;
                double sel_v = 0.0;
                int sel_c = 0;
#pragma unroll
                for (int a = 0; a < AR; ++a) {
                    const bool eq = sc[a] == m;
                    if (eq && !found && pick == 0) {
                        act = a; nfc = c[a].first_child; found = true;
                        sel_v = c[a].value; sel_c = c[a].count;
                    }
                    if (eq && !found) --pick;
                }
#pragma unroll
                for (int i = 0; i < KEEP; ++i)
                    if (depth + 1 == 2 + i) { kv[i] = sel_v; kc[i] = sel_c; }
            } else {
                double m = 0.0;
                for (int a = 0; a < A; ++a) {
                    const UctNode c = tree.child(fc, a);
                    const double sc = c.value + explore(a, c.count + 1);
                    if (a == 0 || sc > m) m = sc;
                }
                int nt = 0;
                for (int a = 0; a < A; ++a) {
                    const UctNode c = tree.child(fc, a);
                    nt += (c.value + explore(a, c.count + 1)) == m ? 1 : 0;
                }
                int pick = (int)g.below((uint32_t)nt);
                for (int a = 0; a < A; ++a) {
                    const UctNode c = tree.child(fc, a);
                    if ((c.value + explore(a, c.count + 1)) == m) {
                        if (pick == 0) { act = a; nfc = c.first_child; break; }
                        --pick;
                    }
                }
            }
            const long idx = (long)s * A + act;
            double reward;
            if (CART) {
                terminal = cartpole_step(p.cp, x4, act, p.cp_sincos, sctab);
                reward = 1.0;
            } else if (LDSM) {
                const uint32_t e = t16[idx];
                reward = LDSR ? rdict[r8[idx]] : rec[idx].reward;
                const bool next_term = (e & 0x8000u) != 0;
                terminal = p.done_on_next ? next_term : cur_term;
                cur_term = next_term;
                s = (int32_t)(e & 0x7fffu);
            } else if (MK) {
                const uint4 q0 = p.pol_frec[(size_t)idx * (1 + NQ32)]; // the (s, a) record with the policy's masks
                terminal = (q0.y & done_bit) != 0;
                reward = __hiloint2double((int)q0.w, (int)q0.z);
                s = (int32_t)q0.x;
                cur_mask = (q0.y >> 8) & 0xffu;
            } else {
                const Rec rc = rec[idx];
                terminal = (rc.flags & done_bit) != 0;
                reward = rc.reward;
                s = rc.next;
            }
            ++st;
            ++steps_taken;
            total += gpow[depth] * reward;
            node = fc + act;
            hnode = tree.handle(fc, act);
            ++depth;
            PATH_PUT(depth, hnode);
            fc = nfc;
#ifdef MP_PROFILE
            ++n_sel;
#endif
        }
        PROF_T(c1);
        // ---- expansion, mcts.py:151-154 / 237-246
        if (owner && fc < 0 && depth < H && (!terminal || node == 0)) {
            UctNode n;
            n.value = 0.0; n.count = 0; n.first_child = -1;
            if (RC && node <= A) {
                if (node == 0) tf0 = n_nodes;
#pragma unroll
                for (int a = 0; a < AR; ++a) tf[a] = node == 1 + a ? n_nodes : tf[a];
            } else {
                tree.at(hnode).first_child = n_nodes;
            }
            if (MK) {
                if (node == 0) {
#pragma unroll
                    for (int a = 0; a < AR; ++a) tc[a] = (cur_mask >> a) & 1u ? 0 : -1;
                } else {
                    UctNode *grp = &tree.child(n_nodes, 0);
                    for (int a = 0; a < A; ++a) {
                        n.count = (cur_mask >> a) & 1u ? 0 : -1;
                        tree.sibling(grp, a) = n;
                    }
                }
            } else if (!(RC && node == 0)) { // the root's children were zero-initialised in registers
                UctNode *grp = &tree.child(n_nodes, 0);
                if (AT > 0) {
#pragma unroll
                    for (int a = 0; a < AR; ++a) tree.sibling(grp, a) = n;
                } else {
                    for (int a = 0; a < A; ++a) tree.sibling(grp, a) = n;
                }
            }
            n_nodes += A;
        }
        PROF_T(c2);
        // ---- rollout, mcts.py:156-157 / 160-177.  Two steps per trip over two generator copies (ga, gb):
        // step "a" consumes the draw in `u`, advances gb = next(ga) speculatively while its lookup is in flight
        // and commits it only if the rollout continues; step "b" mirrors it.  No register copies per step,
        // and a rollout that stops leaves the stream exactly where the reference's would be.  In the LDS
        // variant the reward of a step (HBM/L2, off the state chain) is added one step later, in order.
        if constexpr (QD) {
            // ---- rollout, four lanes per root (see the comment on QD above)
            const bool want = owner && !terminal && depth < H;
            if (__any(want ? 1 : 0)) {
                auto bcastq = [](uint32_t v, int q) { // lane q of the quad, to its four lanes (quad_perm q,q,q,q)
                    return (uint32_t)(q == 0 ? __builtin_amdgcn_mov_dpp((int)v, 0x00, 0xf, 0xf, true)
                                             : (q == 1 ? __builtin_amdgcn_mov_dpp((int)v, 0x55, 0xf, 0xf, true)
                                                       : (q == 2 ? __builtin_amdgcn_mov_dpp((int)v, 0xaa, 0xf, 0xf, true)
                                                                 : __builtin_amdgcn_mov_dpp((int)v, 0xff, 0xf, 0xf, true))));
                };
                // the owner's generator state, to the four lanes of its quad; lane j jumps j + 1 steps ahead (draw k of the
                // rollout is the output of the state after k + 1 steps) -- one table-driven jump, not j + 1 steps
                Pcg64 q = g;
                q.s_lo = (uint64_t)bcastq((uint32_t)g.s_lo, 0) | ((uint64_t)bcastq((uint32_t)(g.s_lo >> 32), 0) << 32);
                q.s_hi = (uint64_t)bcastq((uint32_t)g.s_hi, 0) | ((uint64_t)bcastq((uint32_t)(g.s_hi >> 32), 0) << 32);
                const uint64_t st_lo = q.s_lo, st_hi = q.s_hi;      // (the state the final jump by n starts from)
                {
                    const int j1 = (lane & 3) + 1;
                    uint32_t an[4], gn[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { an[i] = jump[j1 * 8 + i]; gn[i] = jump[j1 * 8 + 4 + i]; }
                    q.jump(an, gn);
                }
                auto draw = [&](const Pcg64 &gen) -> uint32_t {     // searchsorted(cdf, u, 'right') on the raw 64-bit output
                    const uint64_t u = gen.output();
                    int act = 0;
#pragma unroll
                    for (int a = 0; a < NTH; ++a) act += p.thr_arg[a] <= u ? 1 : 0;
                    return (uint32_t)min(act, p.thr_valid);
                };
                // ROUNDS of four draws (one per lane) and four steps of the owner's walk, as long as any root of the wave is
                // still rolling; the draws of round r + 1 are computed beside the walk of round r (independent instruction
                // streams of one basic block: the walk's LDS round trips hide under the generator's multiplies).  Every step is
                // unconditional arithmetic with selects -- a finished root re-reads a valid record and keeps its values.
                // (round 6, as uct_row_kernel's walk) a round in two passes: the state chain alone -- index -> LDS read -> mask, SPECULATIVE
                // past the rollout's end (valid records nobody uses) -- with the rewards' reads beside it; then the round's bookkeeping as
                // bit arithmetic: stop bit j = "the rollout ends AFTER step j", steps taken = first stop bit + 1 (or 4); a step not taken
                // adds +0.0 to a return that is never -0.  The step counters advance once, after the rollout.
                bool alive = want;
                int h = depth;
                const int hmax = p.max_steps > 0 ? min(H, depth + p.max_steps - st) : H;   // (the first step is unconditional)
                uint32_t act_cur = draw(q);
                q.advance4(g4_lo, g4_hi);
                while (__any(alive ? 1 : 0)) {
                    const uint32_t a4[4] = {bcastq(act_cur, 0), bcastq(act_cur, 1), bcastq(act_cur, 2), bcastq(act_cur, 3)};
                    const uint32_t act_next = draw(q);
                    q.advance4(g4_lo, g4_hi);
                    uint32_t e4[4];
                    double rw4[4];
                    int32_t sw = s;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned idx = (unsigned)(sw * A) + a4[i];
                        e4[i] = t16[idx];
                        rw4[i] = rdict[r8[idx]];
                        sw = (int32_t)(e4[i] & 0x7fffu);
                    }
                    const uint32_t nt = ((e4[0] >> 15) & 1u) | ((e4[1] >> 14) & 2u) | ((e4[2] >> 13) & 4u) | ((e4[3] >> 12) & 8u);
                    const uint32_t th = p.done_on_next ? nt : (((nt << 1) | (cur_term ? 1u : 0u)) & 15u);
                    const int left = hmax - h;                      // steps still allowed (the first one always is)
                    const uint32_t lim = left <= 4 ? 1u << ((max(left, 1) - 1) & 3) : 0u;
                    const uint32_t stop = th | lim;
                    const int first = stop ? __ffs((int)stop) - 1 : 4;
                    const int k = alive ? min(first + 1, 4) : 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const double add = gpow[min(h + j, H)] * rw4[j];
                        total += j < k ? add : 0.0;
                    }
                    const uint32_t ek = k <= 1 ? e4[0] : (k == 2 ? e4[1] : (k == 3 ? e4[2] : e4[3]));
                    s = k > 0 ? (int32_t)(ek & 0x7fffu) : s;
                    cur_term = k > 0 ? (ek & 0x8000u) != 0 : cur_term;
                    h += k;
                    alive = alive && first >= 4;
                    act_cur = act_next;
                }
                const int n = h - depth;
                st += n; steps_taken += n;
                if (want) { // the generator after the n draws the walk consumed: A^n state + inc G_n
                    uint32_t an[4], gn[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { an[i] = jump[n * 8 + i]; gn[i] = jump[n * 8 + 4 + i]; }
                    g.s_lo = st_lo; g.s_hi = st_hi;
                    g.jump(an, gn);
                }
            }
        } else
        if (CART && cart_flat && p.rep_shift >= 2) {
            // ---- CartPole rollout, REPLICATED ROOTS (round 6).  The launch code gives a root 2^rep_shift >= 4 lanes that all compute
            // the same plan (why: see there).  What the replicas can do differently is numpy's generator: the action sequence of a
            // rollout does not depend on the states it visits (uniform policy), so lane j of a quad holds the generator j + 1 steps
            // ahead and jumps by four per round (as the four-lanes-per-root table kernel, QD above): a step takes its force from a
            // quad broadcast, and the 128-bit generator step is executed once per FOUR env steps.  The rollout is software-pipelined
            // as the one-lane form below (sin / cos of the next angle beside this step's accelerations).  The walk consumes n draws; the
            // generator then jumps by exactly n.  The three divisions by the constant total mass take the short exact form (div_by_const; FD below).  Every value is cartpole_step's, operation for operation (same bits: tests/test_gpu_cartpole.py).
            bool alive = !terminal && depth < H;
            if (any64(alive)) {
                const mp_cartpole_params &c = p.cp;
                const double total_mass = c.masspole + c.masscart, polemass_length = c.masspole * c.length;
                const double inv_tm = p.cp_inv_tm;
                // FD: the three divisions by the constant total mass as div_by_const (24 vector instructions fewer per step: 0.336 ->
                // 0.298 ms).  Its numerators stay in {0} U [2^-960, 2^1000] while the pole's angular velocity stays below 2^300 with
                // parameters in [2^-50, 2^50] (host check): |a1| = |force + pml thd^2 sn| <= 2^701 and is 0 or >= ulp(force) / 2;
                // masspole cs^2 >= 2^-52 for a live angle; |thetaacc| <= 2^752 / den_min <= 2^802, so |a3| = |pml thetaacc cs| <=
                // 2^902, and a3 is 0 or >= 2^-901 (thetaacc = num / den with num = g sn - cs temp: 0, or >= 2^-209 when a1 != 0, or
                // = g sn >= 2^-747 when a1 = 0, which needs |sn| >= 2^-697 under the velocity bound).  The rollout tracks max |thd|
                // (one v_max_f64 per step, finished roots included: their values stay finite) and is REDONE with IEEE divisions if
                // the bound failed or the velocity is not a number -- nothing is committed before.
                // RL: draws per round = lanes of a root that hold the generator at consecutive steps -- 4 (a quad) or, with sixteen
                // replicas, 16 (a DPP row: the generator step once per SIXTEEN env steps)
                auto roll = [&](auto fma_tag, auto fd_tag, auto rl_tag, bool alive) -> bool {   // (`alive` by value: a redo starts from the same state)
                constexpr bool FMA_FORM = decltype(fma_tag)::value, FD = decltype(fd_tag)::value;
                constexpr int RL = decltype(rl_tag)::value;
                auto divc = [&](double a) -> double {
                    if constexpr (FD) return div_by_const(a, total_mass, inv_tm);
                    else return a / total_mass;
                };
                const bool want = alive;
                const uint64_t st_lo = g.s_lo, st_hi = g.s_hi;      // (the state the final jump by n starts from)
                Pcg64 q = g;
                {
                    const int j1 = min((lane & (RL - 1)) + 1, H);     // (draws beyond the horizon are never used)
                    uint32_t an[4], gn[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { an[i] = cjump[j1 * 8 + i]; gn[i] = cjump[j1 * 8 + 4 + i]; }
                    q.jump(an, gn);
                }
                auto next_round = [&] { if constexpr (RL == 16) q.advance16(g16_lo, g16_hi); else q.advance4(g4_lo, g4_hi); };
                // the high word of the force this lane's draw selects (+- force_mag differ in the sign bit only)
                const int f_lo = __double2loint(c.force_mag);
                const uint32_t f_pos = (uint32_t)__double2hiint(c.force_mag), f_neg = (uint32_t)__double2hiint(-c.force_mag);
                auto force_hi = [&](const Pcg64 &gen) -> uint32_t {
                    const uint64_t u = gen.output();
                    int act = 0;
#pragma unroll
                    for (int a = 0; a < NTH; ++a) act += p.thr_arg[a] <= u ? 1 : 0; // scalar operands (RAWU: the raw 64-bit draw)
                    act = min(act, p.thr_valid);
                    return act == 1 ? f_pos : f_neg;
                };
                uint32_t f_cur = force_hi(q);
                next_round();
                int h = depth;
                // the step after which the rollout stops at the latest: the horizon, or the environment's step limit
                const int hmax = p.max_steps > 0 ? min(H, depth + p.max_steps - st) : H;
                double x = x4[0], x_dot = x4[1], theta = x4[2], theta_dot = x4[3];
                double sn, cs;
                libm_sincos_small_flat<FMA_FORM>(theta, &sn, &cs, sctab);
                // BRANCH-FREE steps: a lane whose rollout ended keeps stepping on values nobody reads (only its return, its step
                // count and `alive` are frozen), so a round of four steps is one basic block for the scheduler to interleave.
                double ret = total;
                double vmax = fabs(theta_dot);             // FD: the largest |angular velocity| a division of this rollout saw
                // The same step with its instruction stream ORDERED BY HAND (FMA form of the host libm + the short exact divisions: what
                // an MI355X host runs).  A lone wave issues in order -- a dependent f64 instruction ~10 cycles after its producer, an
                // independent one after ~5 (tools/ilp_rate.hip) -- and the compiler emits the step's chains one after the other
                // (profiles/r06_cartpole.md).  Here the chains are written side by side, one operation of each per group, and an EMPTY
                // asm volatile that "modifies" the group's results ties them together: the next operation of every chain then has to be
                // emitted after it (a scheduling barrier alone does not do it: instruction selection linearises the block chain by chain
                // before the machine scheduler sees the barrier).  The chains: the velocity recurrence (a1 -> temp -> num -> thetaacc),
                // the other divisor (masspole cs^2 / m -> den), sin's Taylor polynomial and the table path of sin / cos of the NEXT
                // angle, then the IEEE division (the compiler's own sequence, spelled out) beside the table path's corrections, then the
                // last short division beside the bookkeeping.  Operation for operation the arithmetic of one_step below (which stays
                // for the other forms): the same bits.
#define MP_T2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define MP_T3(a, b, c_) asm volatile("" : "+v"(a), "+v"(b), "+v"(c_))
#define MP_T4(a, b, c_, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c_), "+v"(d))
#define MP_T5(a, b, c_, d, e) asm volatile("" : "+v"(a), "+v"(b), "+v"(c_), "+v"(d), "+v"(e))
                auto one_step_il = [&](uint32_t f_hi) {
                    typedef LibmConst K;
                    const double m = total_mass, y = inv_tm;
                    const double force = __hiloint2double((int)f_hi, f_lo);
                    double txd = c.tau * x_dot, tthd = c.tau * theta_dot, thd2 = theta_dot * theta_dot, cs2 = cs * cs;
                    MP_T4(txd, tthd, thd2, cs2);
                    double x_n = x + txd, th_n = theta + tthd, t2 = polemass_length * thd2, a2 = c.masspole * cs2, gsn = c.gravity * sn;
                    MP_T5(x_n, th_n, t2, a2, gsn);
                    const double ax = fabs(th_n);
                    double u = K::big + ax, t3 = t2 * sn, qa0 = a2 * y, xxt = th_n * th_n;
                    MP_T4(u, t3, qa0, xxt);
                    double ub = u - K::big, a1 = force + t3, ra0 = __fma_rn(-qa0, m, a2), pt = __fma_rn(xxt, K::s5, K::s4);
                    int i4 = libm_low_word(u) << 2;
                    i4 = i4 < 0 ? 0 : (i4 > MP_SINCOS_ENTRIES - 4 ? MP_SINCOS_ENTRIES - 4 : i4);
                    const double tsn = sctab[i4], tssn = sctab[i4 + 1], tcs = sctab[i4 + 2], tccs = sctab[i4 + 3];
                    const double gp = gpow[h];
                    MP_T4(ub, a1, ra0, pt);
                    double r = ax - ub, q0 = a1 * y, qa1 = __fma_rn(ra0, y, qa0);
                    pt = __fma_rn(xxt, pt, K::s3);
                    MP_T4(r, q0, qa1, pt);
                    double xx = r * r, r0m = __fma_rn(-q0, m, a1), ra1 = __fma_rn(-qa1, m, a2);
                    pt = __fma_rn(xxt, pt, K::s2);
                    MP_T4(xx, r0m, ra1, pt);
                    double rxx = r * xx, pp = __fma_rn(xx, K::sn5, K::sn3), q1 = __fma_rn(r0m, y, q0), a2q = __fma_rn(ra1, y, qa1);
                    pt = __fma_rn(xxt, pt, K::s1);
                    MP_T5(rxx, pp, q1, a2q, pt);
                    double qq = __fma_rn(xx, K::cs6, K::cs4), r1m = __fma_rn(-q1, m, a1), dd = 4.0 / 3.0 - a2q, tt = __fma_rn(th_n, pt, -0.0);
                    const double dxs = th_n <= 0 ? -0.0 : 0.0;
                    MP_T4(qq, r1m, dd, tt);
                    qq = __fma_rn(xx, qq, K::cs2);
                    double s_cos = __fma_rn(rxx, pp, r), temp = __fma_rn(r1m, y, q1), den = c.length * dd;
                    tt = __fma_rn(tt, xxt, 0.0);
                    MP_T5(qq, s_cos, temp, den, tt);
                    double cc = xx * qq, ssi = __fma_rn(rxx, pp, dxs), cst = cs * temp, sin_taylor = th_n + tt;
                    MP_T4(cc, ssi, cst, sin_taylor);
                    double s_sin = r + ssi, num = gsn - cst;
                    MP_T2(s_sin, num);
                    // thetaacc = num / den: v_div_scale x 2, v_rcp, two refinements, quotient, residual, v_div_fmas, v_div_fixup
                    bool fl0, fl1;
                    double d0 = __builtin_amdgcn_div_scale(num, den, false, &fl0);
                    double e1s = __fma_rn(s_sin, tccs, tssn), e1c = __fma_rn(-s_cos, tssn, tccs);
                    MP_T3(d0, e1s, e1c);
                    double d1 = __builtin_amdgcn_div_scale(num, den, true, &fl1);
                    double rr = __builtin_amdgcn_rcp(d0);
                    double e2s = __fma_rn(-cc, tsn, e1s), e2c = __fma_rn(-cc, tcs, e1c);
                    MP_T4(d1, rr, e2s, e2c);
                    double ea = __fma_rn(-d0, rr, 1.0), cor_s = __fma_rn(s_sin, tcs, e2s), cor_c = __fma_rn(-s_cos, tsn, e2c);
                    MP_T3(ea, cor_s, cor_c);
                    rr = __fma_rn(rr, ea, rr);
                    double sin_tab0 = tsn + cor_s, cs_n = tcs + cor_c;
                    MP_T3(rr, sin_tab0, cs_n);
                    double eb = __fma_rn(-d0, rr, 1.0), sin_tab = copysign(sin_tab0, th_n);
                    const bool fell = fabs(x_n) > c.x_threshold || fabs(th_n) > c.theta_threshold;
                    MP_T2(eb, sin_tab);
                    rr = __fma_rn(rr, eb, rr);
                    double sn_n = libm_select(libm_high_abs(th_n) < 0x3e500000u, th_n, libm_select(ax < 0.126, sin_taylor, sin_tab));
                    MP_T2(rr, sn_n);
                    double qd = d1 * rr;
                    ret = libm_select(alive, ret + gp * 1.0, ret);
                    MP_T2(qd, ret);
                    double ec = __fma_rn(-d0, qd, d1);
                    h += alive ? 1 : 0;
                    alive = alive && !(fell || h >= hmax);
                    double qf = __builtin_amdgcn_div_fmas(ec, rr, qd, fl1);
                    const double thetaacc = __builtin_amdgcn_div_fixup(qf, den, num);
                    double a3a = polemass_length * thetaacc, tth = c.tau * thetaacc;
                    MP_T2(a3a, tth);
                    double a3 = a3a * cs, thd_n = theta_dot + tth;
                    MP_T2(a3, thd_n);
                    const double q30 = a3 * y;
                    vmax = fmax(vmax, fabs(thd_n));
                    const double r30 = __fma_rn(-q30, m, a3);
                    const double q31 = __fma_rn(r30, y, q30);
                    const double r31 = __fma_rn(-q31, m, a3);
                    const double q3 = __fma_rn(r31, y, q31);
                    const double xacc = temp - q3;
                    const double txa = c.tau * xacc;
                    x = x_n; theta = th_n;
                    x_dot = x_dot + txa;
                    theta_dot = thd_n;
                    sn = sn_n; cs = cs_n;
                };
#undef MP_T2
#undef MP_T3
#undef MP_T4
#undef MP_T5
                auto one_step = [&](uint32_t f_hi) {
                    const double force = __hiloint2double((int)f_hi, f_lo);
                    // chain 1: the new positions, then the sin / cos the NEXT step needs
                    const double x_n = x + c.tau * x_dot;
                    const double theta_n = theta + c.tau * theta_dot;
                    double sn_n, cs_n;
                    libm_sincos_small_flat<FMA_FORM>(theta_n, &sn_n, &cs_n, sctab);
                    // chain 2: this step's accelerations from sin / cos of the CURRENT angle
                    const double a1 = force + polemass_length * (theta_dot * theta_dot) * sn;
                    const double temp = divc(a1);
                    const double thetaacc = (c.gravity * sn - cs * temp) / (c.length * (4.0 / 3.0 - divc(c.masspole * (cs * cs))));
                    const double xacc = temp - divc(polemass_length * thetaacc * cs);
                    const double gp = gpow[h];
                    // (x < -T || x > T is |x| > T for every T, NaN included: two compares instead of four)
                    const bool fell = fabs(x_n) > c.x_threshold || fabs(theta_n) > c.theta_threshold;
                    x = x_n; theta = theta_n;
                    x_dot = x_dot + c.tau * xacc;
                    theta_dot = theta_dot + c.tau * thetaacc;
                    if constexpr (FD) vmax = fmax(vmax, fabs(theta_dot));
                    sn = sn_n; cs = cs_n;
                    ret = libm_select(alive, ret + gp * 1.0, ret);
                    h += alive ? 1 : 0;
#ifdef MP_PROFILE
                    n_roll += alive ? 1 : 0;
#endif
                    alive = alive && !(fell || h >= hmax);
                };
                auto step = [&](uint32_t f_hi) {
#ifndef MP_CART_NO_IL
                    if constexpr (FMA_FORM && FD) one_step_il(f_hi); else one_step(f_hi);
#else
                    one_step(f_hi);
#endif
                };
                while (true) {
                    const uint32_t f_next = force_hi(q);
                    next_round();
                    if constexpr (RL == 4) {
                        step(quad_bcast<0>(f_cur)); step(quad_bcast<1>(f_cur)); step(quad_bcast<2>(f_cur)); step(quad_bcast<3>(f_cur));
                    } else {
                        // (the test "is every root of the wave done" after every four steps, as with quads)
                        step(row_bcast<0>(f_cur)); step(row_bcast<1>(f_cur)); step(row_bcast<2>(f_cur)); step(row_bcast<3>(f_cur));
                        if (any64(alive)) {
                            step(row_bcast<4>(f_cur)); step(row_bcast<5>(f_cur)); step(row_bcast<6>(f_cur)); step(row_bcast<7>(f_cur));
                            if (any64(alive)) {
                                step(row_bcast<8>(f_cur)); step(row_bcast<9>(f_cur)); step(row_bcast<10>(f_cur)); step(row_bcast<11>(f_cur));
                                if (any64(alive)) {
                                    step(row_bcast<12>(f_cur)); step(row_bcast<13>(f_cur)); step(row_bcast<14>(f_cur)); step(row_bcast<15>(f_cur));
                                }
                            }
                        }
                    }
                    f_cur = f_next;
                    if (!any64(alive)) break;
                }
                if constexpr (FD)
                    if (any64(want && (!(vmax <= 0x1p300) || theta_dot != theta_dot))) return false;   // (the IEEE form runs it again)
                total = ret;
                const int n = h - depth;                   // env steps = draws of this rollout
                st += n; steps_taken += n;
                if (want) { // the generator after the n draws the walk consumed: A^n state + inc G_n
                    uint32_t an[4], gn[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { an[i] = cjump[n * 8 + i]; gn[i] = cjump[n * 8 + 4 + i]; }
                    g.s_lo = st_lo; g.s_hi = st_hi;
                    g.jump(an, gn);
                }
                return true;
                };
                auto run = [&](auto fma_tag, auto rl_tag) {
                    if (!p.cp_fastdiv || !roll(fma_tag, std::true_type{}, rl_tag, alive)) roll(fma_tag, std::false_type{}, rl_tag, alive);
                };
                typedef std::integral_constant<int, 4> R4;
                typedef std::integral_constant<int, 16> R16;
                if (p.cp_sincos == SINCOS_LIBM_FMA) { if (p.rep_shift >= 4) run(std::true_type{}, R16{}); else run(std::true_type{}, R4{}); }
                else { if (p.rep_shift >= 4) run(std::false_type{}, R16{}); else run(std::false_type{}, R4{}); }
            }
        } else
        if (CART && cart_flat) {
            // ---- CartPole rollout, SOFTWARE-PIPELINED (round 6).  A step's new positions depend on the OLD state only
            // (x + tau x_dot, theta + tau theta_dot), so the sin / cos of the NEXT angle can be evaluated while this step's
            // accelerations -- three IEEE divisions on the velocity chain -- are: two independent dependency chains in one basic
            // block, which the scheduler interleaves (the restated sin / cos is branch-free for that: libm_sincos_small_flat).  A
            // lone wave runs at the latency of its chain: the chain per step was sin / cos + the accelerations, it is now the
            // longer of the two.  The next draw is computed beside them as well and kept only if the rollout continues.  The
            // arithmetic of every value is cartpole_step's, operation for operation (same bits: tests/test_gpu_cartpole.py).
            bool alive = !terminal && depth < H;
            if (any64(alive)) {
                const mp_cartpole_params &c = p.cp;
                const double total_mass = c.masspole + c.masscart, polemass_length = c.masspole * c.length;
                // (the form of the host libm's sin / cos is a compile-time parameter of the loop: a run-time test inside it would
                // split the step into several basic blocks)
                auto roll = [&](auto fma_tag) {
                constexpr bool FMA_FORM = decltype(fma_tag)::value;
                int h = depth;
                double x = x4[0], x_dot = x4[1], theta = x4[2], theta_dot = x4[3];
                double sn, cs;
                libm_sincos_small_flat<FMA_FORM>(theta, &sn, &cs, sctab);
                Pcg64 gr = g;
                uint64_t u = gr.next64();
                while (true) {
                    int act = 0;
#pragma unroll
                    for (int a = 0; a < NTH; ++a) act += p.thr_arg[a] <= u ? 1 : 0; // scalar operands (RAWU: the raw 64-bit draw)
                    act = min(act, p.thr_valid);
                    const double force = act == 1 ? c.force_mag : -c.force_mag;
                    // chain 1: the new positions, then the sin / cos the NEXT step needs
                    const double x_n = x + c.tau * x_dot;
                    const double theta_n = theta + c.tau * theta_dot;
                    double sn_n, cs_n;
                    libm_sincos_small_flat<FMA_FORM>(theta_n, &sn_n, &cs_n, sctab);
                    // chain 2: this step's accelerations from sin / cos of the CURRENT angle
                    const double temp = (force + polemass_length * (theta_dot * theta_dot) * sn) / total_mass;
                    const double thetaacc = (c.gravity * sn - cs * temp) / (c.length * (4.0 / 3.0 - c.masspole * (cs * cs) / total_mass));
                    const double xacc = temp - polemass_length * thetaacc * cs / total_mass;
                    // chain 3: the next draw, kept only if the rollout continues
                    Pcg64 gs = gr;
                    const uint64_t un = gs.next64();
                    const double gp = gpow[h];
                    const bool fell = x_n < -c.x_threshold || x_n > c.x_threshold || theta_n < -c.theta_threshold || theta_n > c.theta_threshold;
                    if (alive) {
                        x = x_n; theta = theta_n;
                        x_dot = x_dot + c.tau * xacc;
                        theta_dot = theta_dot + c.tau * thetaacc;
                        sn = sn_n; cs = cs_n;
                        total += gp * 1.0;
                        ++st; ++steps_taken; ++h;
#ifdef MP_PROFILE
                        ++n_roll;
#endif
                        alive = !(fell || (p.max_steps > 0 && st >= p.max_steps) || h >= H);
                        if (alive) { gr = gs; u = un; }
                    }
                    if (!any64(alive)) break;
                }
                if (!terminal && depth < H) g = gr; // the generator whose draw was consumed last
                };
                if (p.cp_sincos == SINCOS_LIBM_FMA) roll(std::true_type{});
                else roll(std::false_type{});
            }
        } else
        if (!terminal && depth < H) {
            int h = depth;
            // the step after which the rollout stops at the latest: the horizon, or the environment's step limit (the first step
            // is unconditional, as in the reference's loop: hmax <= depth stops after it)
            const int hmax = p.max_steps > 0 ? min(H, depth + p.max_steps - st) : H;
            Pcg64 ga = g, gb = g;
            uint64_t u = ga.next64() >> USH; // the 53 random bits of Generator.random() (RAWU: all 64, see thr_arg)
            bool stopped_in_a = true;
            double r_a = 0.0, r_b = 0.0, g_a = 0.0, g_b = 0.0;
            bool have_b = false;
            uint32_t tcur[NTH]; // SP: top 32 bits of the thresholds of the state the rollout is in (saturated)
            if (SP) {
                const uint4 *tr = reinterpret_cast<const uint4 *>(p.pol_thr + (long)s * p.pol_stride);
                const int shift = p.pol_shift;
                const uint64_t sat = p.pol_sat;
                auto coarse = [shift, sat](uint32_t lo, uint32_t hi) { // min(thr >> shift, sat)
                    const uint64_t c = (((uint64_t)hi << 32) | lo) >> shift;
                    return c > sat ? (uint32_t)sat : (uint32_t)c;
                };
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const uint4 v = tr[q];
                    if (2 * q < NTH) tcur[2 * q] = coarse(v.x, v.y);
                    if (2 * q + 1 < NTH) tcur[2 * q + 1] = coarse(v.z, v.w);
                }
            }
            // one rollout step; returns true when the rollout must stop after it
            auto step = [&](double &r_mine, double &g_mine, const double r_prev, const double g_prev, bool add_prev,
                            const Pcg64 &gcur, Pcg64 &gspec, uint64_t &unext) -> bool {
                // searchsorted(cdf, u, 'right') = #{a : cdf[a] <= u} = #{a : thr[a] <= k}
                int act = 0;
                if (SP) {
                    // thr <= k is decided by the top bits unless they are equal: th < kh => thr < (th+1) << shift
                    // <= k;  th > kh => thr >= th << shift > k (also for the saturated th of thr = 2^53)
                    const uint32_t kh = (uint32_t)(u >> p.pol_shift);
                    bool ambiguous = false;
#pragma unroll
                    for (int a = 0; a < NTH; ++a) {
                        act += tcur[a] < kh ? 1 : 0;
                        ambiguous |= tcur[a] == kh;
                    }
                    if (ambiguous) {
                        const uint64_t *tx = p.pol_thr + (long)s * p.pol_stride;
                        act = 0;
#pragma unroll
                        for (int a = 0; a < NTH; ++a) act += tx[a] <= u ? 1 : 0;
                    }
                } else if (AT > 0) {
                    // (the last threshold, 2^53, is never reached; thresholds of 2^53 before it -- trailing zero
                    // probabilities -- are ~0 here and form a suffix: thr_valid caps the count so that even the draw
                    // 2^64 - 1 selects what searchsorted selects)
#pragma unroll
                    for (int a = 0; a < NTH; ++a) act += p.thr_arg[a] <= u ? 1 : 0; // scalar operands
                    act = min(act, p.thr_valid);
                } else {
                    for (int a = 0; a < A; ++a) act += thr[a] <= u ? 1 : 0;
                }
                g_mine = gpow[h];
                // 32-bit record index (S * |A| < 2^31): one multiply-add instead of a 64-bit address chain
                const unsigned ridx = (unsigned)(s * A + act);
                bool term_h;
                if (CART) {
                    gspec = gcur;
                    unext = gspec.next64() >> USH;
                    term_h = cartpole_step(p.cp, x4, act, p.cp_sincos, sctab);
                    total += g_mine * 1.0;
                } else if (LDSM) {
                    const unsigned idx = ridx;
                    const uint32_t e = t16[idx];
                    r_mine = LDSR ? rdict[r8[idx]] : rec[idx].reward;
                    gspec = gcur;
                    unext = gspec.next64() >> USH;
                    // (the first step of a rollout has no previous reward: r_prev = g_prev = 0.0 then, and adding +0.0 changes no bit
                    // of a return that is never -0 -- no select on `add_prev`)
                    total += g_prev * r_prev;
                    const bool next_term = (e & 0x8000u) != 0;
                    term_h = p.done_on_next ? next_term : cur_term;
                    cur_term = next_term;
                    s = (int32_t)(e & 0x7fffu);
                } else if (SP && NTH <= 4 && p.pol_packed) {
                    // one 16-byte record: the 10 top bits of each threshold ride in the spare bits of next / flags
                    // (an ambiguous draw -- ~0.4 % of the steps -- fetches the exact row, as above)
                    const uint4 q0 = p.pol_frec[ridx];
                    gspec = gcur;
                    unext = gspec.next64() >> USH; // overlaps the gather
                    term_h = (q0.y & done_bit) != 0;
                    s = (int32_t)(q0.x & 0xfffffu);
                    total += g_mine * __hiloint2double((int)q0.w, (int)q0.z);
                    tcur[0] = (q0.x >> 20) & 1023u;
                    if (NTH > 1) tcur[NTH > 1 ? 1 : 0] = (q0.y >> 2) & 1023u;
                    if (NTH > 2) tcur[NTH > 2 ? 2 : 0] = (q0.y >> 12) & 1023u;
                    if (NTH > 3) tcur[NTH > 3 ? 3 : 0] = (q0.y >> 22) & 1023u;
                } else if (SP) {
                    const uint4 *fr = p.pol_frec_roll + (size_t)ridx * (1 + NQ32);
                    const uint4 q0 = fr[0];
                    uint4 qt[NQ32];
#pragma unroll
                    for (int q = 0; q < NQ32; ++q) qt[q] = fr[1 + q];
                    gspec = gcur;
                    unext = gspec.next64() >> USH; // overlaps the gather
                    term_h = (q0.y & done_bit) != 0;
                    s = (int32_t)q0.x;
                    total += g_mine * __hiloint2double((int)q0.w, (int)q0.z);
#pragma unroll
                    for (int q = 0; q < NQ32; ++q) {
                        if (4 * q < NTH) tcur[4 * q] = qt[q].x;
                        if (4 * q + 1 < NTH) tcur[4 * q + 1] = qt[q].y;
                        if (4 * q + 2 < NTH) tcur[4 * q + 2] = qt[q].z;
                        if (4 * q + 3 < NTH) tcur[4 * q + 3] = qt[q].w;
                    }
                } else {
                    const Rec rc = rec[ridx];
                    gspec = gcur;
                    unext = gspec.next64() >> USH; // overlaps the gather
                    term_h = (rc.flags & done_bit) != 0;
                    s = rc.next;
                    total += g_mine * rc.reward;
                }
                // (st and steps_taken advance with h: added once after the rollout; the env's step limit is folded into hmax --
                // three vector instructions fewer in a step that issues ~50)
                ++h;
#ifdef MP_PROFILE
                ++n_roll;
#endif
                return term_h || h >= hmax;
            };
            while (true) {
                uint64_t un;
                if (step(r_a, g_a, r_b, g_b, have_b, ga, gb, un)) { stopped_in_a = true; break; }
                u = un;
                have_b = true;
                if (step(r_b, g_b, r_a, g_a, true, gb, ga, un)) { stopped_in_a = false; break; }
                u = un;
            }
            if (LDSM) total += stopped_in_a ? g_a * r_a : g_b * r_b;
            g = stopped_in_a ? ga : gb; // the generator whose draw was consumed last
            st += h - depth; steps_taken += h - depth;
        }
        PROF_T(c3);
        // ---- backup, mcts.py:248-265: the same return for every node on the path
        for (int d = depth; d >= (RC ? 2 : 0); --d) {
            int n;
            PATH_GET(n, d);
            if (RC && d < 2 + KEEP) {
                double v = kv[0];
                int cnt = kc[0];
#pragma unroll
                for (int i = 1; i < KEEP; ++i) { v = d == 2 + i ? kv[i] : v; cnt = d == 2 + i ? kc[i] : cnt; }
                cnt += 1;
                v += inv(cnt) * (total - v);
                tree.at(n).value = v;       // first_child is left alone (the node may just have been expanded)
                tree.at(n).count = cnt;
            } else {
                UctNode c = tree.at(n);
                c.count += 1;
                c.value += inv(c.count) * (total - c.value);
                tree.at(n).value = c.value;
                tree.at(n).count = c.count;
            }
        }
        if (RC) {
            if (depth >= 1) {
                int n;
                PATH_GET(n, 1);
                n -= tree.handle(1, 0); // level-1 node: which child of the root
                double v = tv[0];
                int cnt = tc[0];
#pragma unroll
                for (int a = 1; a < AR; ++a) { v = n == a ? tv[a] : v; cnt = n == a ? tc[a] : cnt; }
                cnt += 1;
                v += inv(cnt) * (total - v);
#pragma unroll
                for (int a = 0; a < AR; ++a) { tv[a] = n == a ? v : tv[a]; tc[a] = n == a ? cnt : tc[a]; }
            }
            tc0 += 1;
            tv0 += inv(tc0) * (total - tv0);
        }
#ifdef MP_PROFILE
        { const long long c4 = clock64(); t_sel += c1 - c0; t_expd += c2 - c1; t_roll += c3 - c2; t_bak += c4 - c3; }
#endif
    }
#ifdef MP_PROFILE
    if (r == 0)
        printf("uct prof wave0: total=%lld select=%lld expand=%lld rollout=%lld backup=%lld | lane0 select steps=%lld rollout steps=%lld\n",
               (long long)(clock64() - t_all0), t_sel, t_expd, t_roll, t_bak, n_sel, n_roll);
#endif
    if (!owner) return;                      // (QD: the helper lanes held no tree)
    if (CART && (lane & ((1 << p.rep_shift) - 1)) != 0) return;   // (replicas of a CartPole root: the first lane reports)
    if (RC) {
        UctNode w;
        w.value = tv0; w.count = tc0; w.first_child = tf0;
        tree[0] = w;
        if (tf0 == 1) {
#pragma unroll
            for (int a = 0; a < AR; ++a) {
                w.value = tv[a]; w.count = tc[a]; w.first_child = tf[a];
                tree.child(1, a) = w;
            }
        }
    }
    g.store(p.rng + (long)r * 6);
    // ---- AbstractPlanner.get_plan (abstract.py:143-156) with MCTSNode.selection_rule
    // (mcts.py:212-218): most visited child, ties -> first maximal value among them
    {
        int len = 0;
        int fc = tree[0].first_child;
        while (fc >= 0) {
            int mc = tree.child(fc, 0).count;
            for (int a = 1; a < A; ++a) mc = max(mc, tree.child(fc, a).count);
            int best = -1;
            double bv = 0.0;
            for (int a = 0; a < A; ++a) {
                const UctNode c = tree.child(fc, a);
                if (c.count == mc && (best < 0 || c.value > bv)) { best = a; bv = c.value; }
            }
            if (p.plans && len < p.max_plan_len) p.plans[(long)r * p.max_plan_len + len] = best;
            ++len;
            fc = tree.child(fc, best).first_child;
        }
        if (p.plans)
            for (int i = len; i < p.max_plan_len; ++i) p.plans[(long)r * p.max_plan_len + i] = -1;
        if (p.plan_len) p.plan_len[r] = len;
    }
    if (p.n_nodes_out) p.n_nodes_out[r] = n_nodes;
    if (p.root_value) p.root_value[r] = tree[0].value;
    if (p.env_steps) p.env_steps[r] = (int64_t)steps_taken;
    const int rfc = tree[0].first_child;
    for (int a = 0; a < A; ++a) {
        if (p.root_child_count) p.root_child_count[(long)r * A + a] = rfc >= 0 ? max(tree.child(rfc, a).count, 0) : 0;
        if (p.root_child_value) p.root_child_value[(long)r * A + a] = rfc >= 0 ? tree.child(rfc, a).value : 0.0;
    }
}

// ---- ONE ROOT PER WORKGROUP (round 5): the plan of a single agent's act(), and of every batch small enough to give each root
// a CU of its own.  A lone root is one dependency chain; uct_kernel runs it in one lane of a wave, the tree in global memory
// (a dependent L2 round trip per scored level and per backed-up node), every rollout step behind numpy's 128-bit generator
// step.  Here the whole WAVEFRONT works for the root:
//   * the model (3 B per (s, a)), the per-call tables AND THE TREE live in LDS (a 33-episode tree is 2.7 KB); sixteen waves
//     stage them, one stays;
//   * a level's |A| children are scored by |A| lanes, a path's nodes are backed up by as many lanes at once;
//   * a rollout's actions do not depend on the states it visits (state-independent policy): lane l draws action l from the
//     generator l + 1 steps ahead (one table-driven jump: A^(l+1) s + inc G_(l+1)), then the walk is H dependent LDS reads
//     and nothing else -- the rewards are looked up by the lanes in parallel afterwards and added in the reference's order;
//     the generator then jumps by the n draws the walk consumed, so the stream stays the reference's.
// Same trees, plans, statistics and generator records as uct_kernel (every single-root golden runs through it when forced:
// MP_UCT_LONE=1); the tree is written to global memory in the group-interleaved layout at the end (export, re-rooting).
// EACH (round 6): ONE MDP PER ROOT -- a batch model (mp_model_load_table_batch: the finite MDPs of a batch of episodes, each
// extracted from its own environment) where root r plans on MDP root_state[r] / Sb.  The union model is too large for the compact
// LDS forms (S = NB * Sb >= 32 768 states), but a root only ever visits ITS MDP: the workgroup stages that MDP's Sb * |A| records
// from the 16-byte global records into LDS as {uint16 local next state | bit 15 terminal[next]} + the reward itself (f64: no
// reward dictionary, any number of distinct rewards) -- 10 B per (s, a), 6 KB for a 120-state highway grid -- and everything
// else is the lone-root kernel: tree in LDS, a level scored across lanes, the rollout's draws by jump-ahead in the lanes.  One
// wave per workgroup (small MDPs) so that 4096 roots are 4096 wavefronts, four per SIMD.
// MW (round 6): SEVERAL ROOTS PER WORKGROUP, A WAVEFRONT EACH, around ONE copy of a model that fills the LDS -- the headline table's
// transitions are 100 of the CU's 160 KB, so with the reward indices beside them only one root fits a CU (256 roots per launch).
// Here the workgroup keeps the transitions only, p.waves (2 / 4 / 8) planning wavefronts stay after the staging, each with its own
// tree and its own copy of the jump table behind them, one or two per SIMD; an episode's rewards come from the model's 16-byte
// records in L2 (ONE vector load for the whole episode: lane l its step's reward) instead of the two LDS round trips.
template <int AT, bool EACH = false, bool MW = false>
__global__ __launch_bounds__(EACH ? 256 : 1024, EACH ? 4 : 1) void uct_lone_kernel(UctArgs p)
{
    static_assert(!(EACH && MW), "one MDP per root is staged per workgroup: one root each");
    static_assert(AT >= 2 && AT <= 8, "|A| with a compile-time specialisation");
    constexpr int A = AT, NTH = AT - 1;
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int H = p.horizon, E = p.episodes, TE = p.table_n;
    double *gpow = lds_d;                   // [H + 1]  gamma ** h
    double *tp = gpow + (H + 1) + A;        // [A]      temperature * |A| * prior[a]
    double *rcp = tp + A;                   // [TE + 1]  1.0 / n
    double *tpdiv = rcp + (TE + 1);         // [A][TE+2] temperature * |A| * prior[a] / n
    const int ntab = (H + 1) + 2 * A + (TE + 1) + A * (TE + 2);
    const int ntab2 = (ntab + 1) & ~1;
    const int SA = (EACH ? p.Sb : p.S) * A;                                       // (s, a) pairs of the model this root plans on
    const double *rdict = lds_d + ntab2;                                          // shared model: [n_rdict] distinct rewards (not MW)
    double *rew = lds_d + ntab2;                                                  // EACH: [SA] the rewards themselves
    uint16_t *t16 = EACH ? reinterpret_cast<uint16_t *>(rew + ((SA + 1) & ~1))
                         : reinterpret_cast<uint16_t *>(lds_d + ntab2 + (MW ? 0 : ((p.n_rdict + 1) & ~1)));
    uint8_t *r8 = reinterpret_cast<uint8_t *>(t16 + ((SA + 7) & ~7));             // shared model: [SA] reward indices (not MW)
    // per planning wavefront: its copy of the jump table [H + 5][8] (it gets the root's inc G_j), its tree [cap] and a node's
    // exploration term at its count [cap]
    uint32_t *jump0 = (EACH || MW) ? reinterpret_cast<uint32_t *>(r8) : reinterpret_cast<uint32_t *>(r8 + ((SA + 15) & ~15));
    const int W = MW ? p.waves : 1;
    const int wstride = ((H + 5) * 8 + p.cap * 6 + 3) & ~3;                       // 32-bit words per wavefront (16-byte multiples)
    uint32_t *jump = jump0 + (MW ? (tid >> 6 < W ? tid >> 6 : 0) * wstride : 0);
    UctNode *tnode = reinterpret_cast<UctNode *>(jump + (H + 5) * 8);            // [cap]
    double *texpl = reinterpret_cast<double *>(tnode + p.cap);                   // [cap]
    // (the [H + 1] words behind texpl held the path until round 6: it now lives in a register, one node per lane)
    int r = blockIdx.x;
    int32_t s0g = MW ? 0 : __builtin_amdgcn_readfirstlane(p.root_state[r]);      // (global state of a batch model)
    const int32_t sbase = EACH ? (s0g / p.Sb) * p.Sb : 0;                         // first global state of this root's MDP
    for (int w = 0; w < W; ++w)
        for (int i = tid; i < (H + 5) * 8; i += nthreads) jump0[w * wstride + i] = p.jump[i];
    for (int i = tid; i < ntab; i += nthreads) lds_d[i] = p.tab[i];
    if constexpr (EACH) {
        const Rec *src = p.rec + (long)sbase * A;
        for (int i = tid; i < SA; i += nthreads) {
            const Rec rc = src[i];
            t16[i] = (uint16_t)((uint32_t)(rc.next - sbase) | ((rc.flags & 2u) ? 0x8000u : 0u));
            rew[i] = rc.reward;
        }
    } else {
        if constexpr (!MW) {
            for (int i = tid; i < p.n_rdict; i += nthreads) lds_d[ntab2 + i] = p.rdict[i];
            const int n16 = (p.S * A + 15) >> 4; // (the device arrays are padded to whole 16-byte chunks)
            const uint4 *src = reinterpret_cast<const uint4 *>(p.r8);
            uint4 *dst = reinterpret_cast<uint4 *>(r8);
            for (int i = tid; i < n16; i += nthreads) dst[i] = src[i];
        }
        const int n = p.S * A, n8 = n >> 3;
        const uint4 *src2 = reinterpret_cast<const uint4 *>(p.t16);
        uint4 *dst2 = reinterpret_cast<uint4 *>(t16);
        for (int i = tid; i < n8; i += nthreads) dst2[i] = src2[i];
        for (int i = (n8 << 3) + tid; i < n; i += nthreads) t16[i] = p.t16[i];
    }
    __syncthreads();
    if (tid >= 64 * W) return; // (the staging waves are done; no barrier below)
    if constexpr (MW) {
        r = blockIdx.x * W + (tid >> 6);
        if (r >= p.n_roots) return;
        s0g = __builtin_amdgcn_readfirstlane(p.root_state[r]);
    }
    // MCTSNode.selection_strategy's exploration term temperature |A| prior[a] / (count + 1) (mcts.py:275-286): from the host's
    // quotient table, or the same IEEE division beyond it.  Kept PER NODE beside the tree and refreshed by the backup (which
    // changes the count), so that scoring a level is one LDS round trip, not two
    auto explore = [&](int a, int cnt1) { return cnt1 <= TE + 1 ? tpdiv[a * (TE + 2) + cnt1] : tp[a] / (double)cnt1; };
    auto inv = [&](int c) { return c <= TE ? rcp[c] : 1.0 / (double)c; };
    Pcg64U g;                                    // (one generator for the wave, in scalar registers)
    g.load(p.rng + (long)r * 6);
    const int32_t s0 = s0g - sbase;                                    // (local to the staged MDP)
    const int32_t st0 = p.root_steps ? __builtin_amdgcn_readfirstlane(p.root_steps[r]) : 0;
    const bool root_term = (p.rec[(long)s0g * A].flags & 1u) != 0; // terminal flag of the root state itself ("source" rule)
    int n_nodes = 1, steps_taken = 0;
    if (lane == 0) { UctNode n; n.value = 0.0; n.count = 0; n.first_child = -1; tnode[0] = n; texpl[0] = 0.0; } // mcts.py:129-130 reset()
    // A jump by j steps is state <- A^j state + inc G_j; the increment is the root's for the whole plan, so this workgroup's copy of
    // the jump table gets inc G_j in place of G_j once (lane l: j = l + 1; the draws use j <= H <= 63) -- every rollout's draw is then
    // ONE 128-bit product per lane, not two
    if (lane < H) {
        const int j = lane + 1;
        uint32_t gn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) gn[i] = jump[j * 8 + 4 + i];
        uint64_t lo, hi;
        Pcg64::mul128v(g.inc_lo, g.inc_hi, gn, lo, hi);
        jump[j * 8 + 4] = (uint32_t)lo; jump[j * 8 + 5] = (uint32_t)(lo >> 32);
        jump[j * 8 + 6] = (uint32_t)hi; jump[j * 8 + 7] = (uint32_t)(hi >> 32);
    }
    __builtin_amdgcn_wave_barrier();
    const int la = lane < A ? lane : 0;
    const double expl1 = explore(la, 1);          // a fresh child's term (count 0)
    // (round 6, second pass) An env step's reward comes through two more LDS reads (reward index -> reward) that the state chain
    // does not need.  A lone wave issues an instruction every ~5 cycles whatever it depends on, so what a step costs is its
    // instruction COUNT: the steps of an episode (descent and rollout alike) only note their (s, a) index in lane `depth` of one
    // register; after the walk the lanes look up their steps' rewards AT ONCE (two LDS reads for the whole episode), multiply by
    // gamma ** lane, and the products are added one by one in the reference's order (mcts.py:160-177: total += gamma ** h * r).
    // The three-stage reward pipeline this replaces cost ~25 of a rollout step's 43 instructions.
    const double gpl = gpow[min(lane, H)];        // gamma ** lane
    int idxv = 0;                                 // lane l: s * |A| + a of the episode's step l (stale beyond: valid, unused)
    uint32_t sv = 0;                              // lane l: the state the rollout's step at depth l leaves from (noted by the walk)
    int pathv = 0;                                // lane l: the tree node at depth l of the episode's path (lane 0: the root)
    const uint32_t tshift = p.done_on_next ? 0u : 16u;
    const uint32_t t16_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t *)t16; // byte offset of t16 in LDS
#ifdef MP_PROFILE
    long long pt[6] = {0, 0, 0, 0, 0, 0}, pn_sel = 0, pn_roll = 0;
    const long long pt_all = clock64();
#define MP_LT(i_) do { const long long c_ = clock64(); pt[i_] += c_ - pc; pc = c_; } while (0)
#else
#define MP_LT(i_) do { } while (0)
#endif
    for (int ep = 0; ep < E; ++ep) { // mcts.py:179-184
#ifdef MP_PROFILE
        long long pc = clock64();
#endif
        int32_t s = s0, st = st0;
        int node = 0, depth = 0, n_roll = 0;
        bool terminal = false;
        // the records of the last two steps, the newer in the low half: bit 15 of the one the terminal rule names ends the descent
        // ("next" rule: the step just taken; "source" rule: the one before, i.e. the state the step left from was terminal)
        uint32_t hist = root_term ? 0x8000u : 0u;
        int fc = __builtin_amdgcn_readfirstlane(tnode[0].first_child);
        // ---- selection, mcts.py:143-149: a level's children one per lane
        while (depth < H && fc >= 0 && !terminal) {
            const UctNode c = tnode[fc + la];
            const double ex = texpl[fc + la];
            // (every child's transition is read with the children, one per lane: the step of the picked one is then a readlane,
            // not a second LDS round trip behind the pick)
            const uint32_t cand = t16[s * A + la];
            const double sc = c.value + ex;
            // Node.random_argmax, abstract.py:296-311: the maxima = the children no other child beats (|A| independent compares
            // against the scores read off their lanes; a chain of maxima would be |A| dependent selects)
            unsigned long long ties = (1ull << A) - 1ull;
#pragma unroll
            for (int a = 0; a < A; ++a) ties &= ballot64(sc >= bcast_lane(sc, a));
            const int nt = __popcll(ties);
            int pick = nt > 1 ? (int)g.below((uint32_t)nt) : 0;
            pick = __builtin_amdgcn_readfirstlane(pick);
            unsigned long long t = ties;
            while (pick-- > 0) t &= t - 1;
            const int act = __ffsll((long long)t) - 1;
            const int nfc = __builtin_amdgcn_readlane(c.first_child, act);
            const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)cand, act);
            idxv = lane == depth ? s * A + act : idxv;
            hist = (hist << 16) | e;
            terminal = ((hist >> tshift) & 0x8000u) != 0;
            s = (int32_t)(e & 0x7fffu);
            ++st; ++steps_taken;
            node = fc + act;
            ++depth;
            pathv = lane == depth ? node : pathv;
            fc = nfc;
        }
        const bool cur_term = (hist & 0x8000u) != 0;
#ifdef MP_PROFILE
        pn_sel += depth;
#endif
        MP_LT(0);
        // ---- expansion, mcts.py:151-154 / 237-246
        if (fc < 0 && depth < H && (!terminal || node == 0)) {
            if (lane == 0) tnode[node].first_child = n_nodes;
            if (lane < A) {
                UctNode n;
                n.value = 0.0; n.count = 0; n.first_child = -1;
                tnode[n_nodes + lane] = n;
                texpl[n_nodes + lane] = expl1;
            }
            n_nodes += A;
        }
        MP_LT(1);
        // ---- rollout, mcts.py:156-157 / 160-177
        if (!terminal && depth < H) {
            // Lane h holds what the step at depth h needs: its action (lane h draws action h - depth of the rollout, from the
            // generator h - depth + 1 steps ahead) and, noted by the walk, the state it leaves from -- so the lanes form their
            // steps' (s, a) indices themselves afterwards.
            uint32_t act_l;
            Pcg64 q;                               // (the lanes' jumps: vector arithmetic on copies)
            q.s_hi = g.s_hi; q.s_lo = g.s_lo; q.inc_hi = g.inc_hi; q.inc_lo = g.inc_lo; q.has_uint32 = 0; q.uinteger = 0;
            {
                const int j1 = min(max(lane - depth + 1, 1), H); // (draws beyond the horizon / of the descent's lanes are never used)
                uint32_t an[4], ig[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { an[i] = jump[j1 * 8 + i]; ig[i] = jump[j1 * 8 + 4 + i]; }
                uint64_t p_lo, p_hi;                // A^j state + inc G_j, the second product from this root's table
                Pcg64::mul128v(q.s_lo, q.s_hi, an, p_lo, p_hi);
                const uint64_t q_lo = (uint64_t)ig[0] | ((uint64_t)ig[1] << 32), q_hi = (uint64_t)ig[2] | ((uint64_t)ig[3] << 32);
                const uint64_t lo = p_lo + q_lo;
                q.s_hi = p_hi + q_hi + (lo < p_lo ? 1ULL : 0ULL);
                q.s_lo = lo;
                const uint64_t u = q.output();     // searchsorted(cdf, u, 'right') on the raw 64-bit output
                int act = 0;
#pragma unroll
                for (int a = 0; a < NTH; ++a) act += p.thr_arg[a] <= u ? 1 : 0;
                act_l = (uint32_t)min(act, p.thr_valid);
            }
            MP_LT(2);
            // the walk: at most n_lim steps (horizon, the env's step limit).  The chain of a step is ONE multiply-add (LDS byte
            // address of the record = state * 2|A| + (2 a + the table's offset), the second term a scalar read off the action's lane)
            // and the read itself; the scalar side only tests the terminal bit and counts.
            int n_lim = H - depth;
            if (p.max_steps > 0 && p.max_steps - st < n_lim) n_lim = p.max_steps - st;
            if (n_lim < 1) n_lim = 1; // (the first step is unconditional, as in the reference's loop)
            const uint32_t act2 = 2u * act_l + t16_lds;
            const int h_end = depth + n_lim;
            // The walk, written out (inline assembly: the compiler rotates any C form of it back into read -> wait -> test, and a
            // lone wave pays ~20 cycles for every branch, taken or not -- tools/lds_chain.hip: the bare chain v_and -> v_mad_u32_u24
            // (state * 2|A| + action term) -> ds_read_u16 is 64 cycles a step, with two exit tests beside it 122).
            //  * GROUPS OF FOUR steps without a branch: each slot issues the read of its step (speculative: any record's next state
            //    is a valid state, its lane's action a valid action; what a step not taken leaves behind is never used) and, under
            //    that read's latency, notes the state the step leaves from in its lane, fetches the next action term and shifts the
            //    previous record's terminal bit into a scalar; ONE test per group finds the step the rollout ends with (the first set
            //    bit: "next" rule = the records of the four steps, "source" rule = the records before them).  Groups run while four
            //    more steps are within the horizon / step limit;
            //  * the remaining (< 4) steps one by one, each with its two tests (the same slot otherwise).
            // The walk starts from a VIRTUAL record: the state before the first step, bit 15 = "that state is terminal" under the
            // "source" rule -- the first step is then an ordinary one (unconditional: the reference's loop tests after the step).
            // Wait states by hand (the hazard pass does not look inside): a VALU that reads an SGPR / VCC a VALU wrote comes at
            // least two instructions later.
            uint32_t v1 = (uint32_t)s | (!p.done_on_next && cur_term ? 0x8000u : 0u);
            int h = depth;                              // (v1 = the record of the step at depth h - 1; h <= H <= 63: a lane)
            uint32_t done = 0;
            if (h + 4 <= h_end) {
                uint32_t w1, w2, w3, w4, t, a2, e, acc = 0, tst, h4;
#define MP_LONE_SLOT(WP, WN)                                                                       \
    "v_and_b32 %[t], 0x7fff, " WP "\n\t"                                                          \
    "v_mad_u32_u24 %[t], %[t], %[mul], %[a2]\n\t"                                                  \
    "ds_read_u16 " WN ", %[t]\n\t"                                                                 \
    "v_cmp_eq_u32 vcc, %[h], %[lane]\n\t"                                                          \
    "v_readfirstlane_b32 %[e], " WP "\n\t"                                                         \
    "s_add_i32 %[h], %[h], 1\n\t"                                                                  \
    "v_cndmask_b32 %[sv], %[sv], " WP ", vcc\n\t"                                                  \
    "v_readlane_b32 %[a2], %[act2], %[h]\n\t"                                                      \
    "s_lshr_b32 %[e], %[e], 15\n\t"                                                                \
    "s_lshl1_add_u32 %[acc], %[acc], %[e]\n\t"                                                     \
    "s_waitcnt lgkmcnt(0)\n\t"
#define MP_LONE_GROUPS(TEST)                                                                       \
    asm volatile("s_nop 3\n\t"                                                                     \
                 "v_readlane_b32 %[a2], %[act2], %[h]\n\t"                                          \
                 "s_nop 1\n\t"                                                                      \
                 "1:\n\t"                                                                           \
                 MP_LONE_SLOT("%[v1]", "%[w1]") MP_LONE_SLOT("%[w1]", "%[w2]")                      \
                 MP_LONE_SLOT("%[w2]", "%[w3]") MP_LONE_SLOT("%[w3]", "%[w4]")                      \
                 TEST                                                                               \
                 "s_cbranch_scc1 2f\n\t"                                                            \
                 "v_mov_b32 %[v1], %[w4]\n\t"                                                       \
                 "s_add_i32 %[h4], %[h], 4\n\t"                                                     \
                 "s_cmp_le_u32 %[h4], %[hend]\n\t"                                                  \
                 "s_cbranch_scc1 1b\n\t"                                                            \
                 "s_mov_b32 %[done], 0\n\t"                                                         \
                 "s_branch 3f\n\t"                                                                  \
                 "2:\n\t"               /* the group's step tst names is the last: (leading zeros of tst) - 27 of its 4 were taken */ \
                 "s_flbit_i32_b32 %[tst], %[tst]\n\t"                                               \
                 "s_add_i32 %[h], %[h], %[tst]\n\t"                                                 \
                 "s_sub_i32 %[h], %[h], 31\n\t"                                                     \
                 "s_mov_b32 %[done], 1\n\t"                                                         \
                 "3:\n\t"                                                                           \
                 : [h] "+s"(h), [sv] "+v"(sv), [v1] "+v"(v1), [acc] "+s"(acc), [done] "+s"(done), [w1] "=&v"(w1), [w2] "=&v"(w2), \
                   [w3] "=&v"(w3), [w4] "=&v"(w4), [t] "=&v"(t), [a2] "=&s"(a2), [e] "=&s"(e), [tst] "=&s"(tst), [h4] "=&s"(h4)  \
                 : [act2] "v"(act2), [lane] "v"(lane), [hend] "s"(h_end), [mul] "n"(2 * A)          \
                 : "vcc", "scc", "memory")
                if (p.done_on_next)
                    MP_LONE_GROUPS("v_readfirstlane_b32 %[e], %[w4]\n\t"
                                   "s_lshr_b32 %[e], %[e], 15\n\t"
                                   "s_lshl1_add_u32 %[tst], %[acc], %[e]\n\t"
                                   "s_and_b32 %[tst], %[tst], 15\n\t");
                else
                    MP_LONE_GROUPS("s_and_b32 %[tst], %[acc], 15\n\t");
#undef MP_LONE_GROUPS
#undef MP_LONE_SLOT
            }
            done = (uint32_t)__builtin_amdgcn_readfirstlane((int)done);   // (scalar for the compiler too, whatever it made of the merge)
            h = __builtin_amdgcn_readfirstlane(h);
            if (!done) {
                uint32_t pe = 0;                        // (the record before v1; a set bit 15 would have ended the walk already)
                uint32_t v2, t, a2, e, h1;
#define MP_LONE_HALF(V1, V2, STOP)                                                                 \
    "v_and_b32 %[t], 0x7fff, " V1 "\n\t"                                                          \
    "v_mad_u32_u24 %[t], %[t], %[mul], %[a2]\n\t"                                                  \
    "ds_read_u16 " V2 ", %[t]\n\t"                                                                 \
    "v_cmp_eq_u32 vcc, %[h], %[lane]\n\t"                                                          \
    "v_readfirstlane_b32 %[e], " V1 "\n\t"                                                         \
    "s_add_i32 %[h1], %[h], 1\n\t"                                                                 \
    "v_cndmask_b32 %[sv], %[sv], " V1 ", vcc\n\t"                                                  \
    "v_readlane_b32 %[a2], %[act2], %[h1]\n\t"                                                     \
    STOP                                                                                           \
    "s_cbranch_scc1 2f\n\t"                                                                        \
    "s_cmp_ge_u32 %[h], %[hend]\n\t"                                                               \
    "s_cbranch_scc1 2f\n\t"                                                                        \
    "s_mov_b32 %[h], %[h1]\n\t"                                                                    \
    "s_waitcnt lgkmcnt(0)\n\t"
#define MP_LONE_WALK(STOP)                                                                         \
    asm volatile("s_nop 3\n\t"                                                                     \
                 "v_readlane_b32 %[a2], %[act2], %[h]\n\t"                                          \
                 "s_nop 1\n\t"                                                                      \
                 "1:\n\t" MP_LONE_HALF("%[v1]", "%[v2]", STOP) MP_LONE_HALF("%[v2]", "%[v1]", STOP)  \
                 "s_branch 1b\n\t"                                                                  \
                 "2:\n\t"                                                                           \
                 "s_waitcnt lgkmcnt(0)"                                                             \
                 : [h] "+s"(h), [sv] "+v"(sv), [v1] "+v"(v1), [pe] "+s"(pe), [v2] "=&v"(v2), [t] "=&v"(t), [a2] "=&s"(a2),       \
                   [e] "=&s"(e), [h1] "=&s"(h1)                                                     \
                 : [act2] "v"(act2), [lane] "v"(lane), [hend] "s"(h_end), [mul] "n"(2 * A)          \
                 : "vcc", "scc", "memory")
                if (p.done_on_next) MP_LONE_WALK("s_bitcmp1_b32 %[e], 15\n\t");
                else MP_LONE_WALK("s_bitcmp1_b32 %[pe], 15\n\ts_mov_b32 %[pe], %[e]\n\t");
#undef MP_LONE_WALK
#undef MP_LONE_HALF
            }
            MP_LT(3);
            const int n = h - depth;
            n_roll = n;
            st += n; steps_taken += n;
            idxv = lane >= depth ? (int)((sv & 0x7fffu) * A + act_l) : idxv;
            {   // the generator after the n draws the walk consumed = the state the lane of the last step jumped to for ITS draw
                const int src = h - 1;
                g.s_lo = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(q.s_lo >> 32), src) << 32) |
                         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)q.s_lo, src);
                g.s_hi = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(q.s_hi >> 32), src) << 32) |
                         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)q.s_hi, src);
            }
        }
        MP_LT(2);
        // the episode's return: its L steps' rewards looked up by L lanes, added in step order (a group of four per uniform test)
        double total = 0.0;
        {
            const int L = depth + n_roll;
            double rd;
            if constexpr (EACH) rd = rew[idxv]; else if constexpr (MW) rd = p.rec[idxv].reward; else rd = rdict[r8[idxv]];
            double prod = gpl * rd;
            prod = lane < L ? prod : 0.0;     // (+0.0 leaves a sum that started at +0.0 as it is, bit for bit)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (q * 4 >= L) break;
#pragma unroll
                for (int j = 0; j < 4; ++j) total += bcast_lane(prod, q * 4 + j);
            }
        }
#ifdef MP_PROFILE
        pn_roll += n_roll;
#endif
        MP_LT(4);
        // ---- backup, mcts.py:248-265: the same return for every node on the path, one node per lane
        __builtin_amdgcn_wave_barrier();
        if (lane <= depth) {
            const int nd = pathv;
            UctNode c = tnode[nd];
            c.count += 1;
            c.value += inv(c.count) * (total - c.value);
            tnode[nd].value = c.value;       // (first_child is left alone: the node may just have been expanded)
            tnode[nd].count = c.count;
            if (nd > 0) texpl[nd] = explore((nd - 1) % A, c.count + 1);
        }
        __builtin_amdgcn_wave_barrier();
        MP_LT(5);
    }
#undef MP_LT
#ifdef MP_PROFILE
    if (r == 0 && lane == 0)
        printf("uct_lone prof: total=%lld select=%lld expand=%lld draw+gen=%lld walk=%lld sum=%lld backup=%lld | levels=%lld rollout steps=%lld\n",
               (long long)(clock64() - pt_all), pt[0], pt[1], pt[2], pt[3], pt[4], pt[5], pn_sel, pn_roll);
#endif
    // the tree, to global memory in the group-interleaved layout (export, re-rooting by step_by_subtree)
    const TreeRef<2, AT> tree = tree_of<2, AT>(p.tree, r, p.cap, A);
    for (int i = lane; i < n_nodes; i += 64) tree[i] = tnode[i];
    if (lane == 0) {
        g.store(p.rng + (long)r * 6);
        // ---- AbstractPlanner.get_plan (abstract.py:143-156) with MCTSNode.selection_rule (mcts.py:212-218): most visited
        // child, ties -> first maximal value among them
        int len = 0;
        int fc = tnode[0].first_child;
        while (fc >= 0) {
            int mc = tnode[fc].count;
            for (int a = 1; a < A; ++a) mc = max(mc, tnode[fc + a].count);
            int best = -1;
            double bv = 0.0;
            for (int a = 0; a < A; ++a) {
                const UctNode c = tnode[fc + a];
                if (c.count == mc && (best < 0 || c.value > bv)) { best = a; bv = c.value; }
            }
            if (p.plans && len < p.max_plan_len) p.plans[(long)r * p.max_plan_len + len] = best;
            ++len;
            fc = tnode[fc + best].first_child;
        }
        if (p.plans)
            for (int i = len; i < p.max_plan_len; ++i) p.plans[(long)r * p.max_plan_len + i] = -1;
        if (p.plan_len) p.plan_len[r] = len;
        if (p.n_nodes_out) p.n_nodes_out[r] = n_nodes;
        if (p.root_value) p.root_value[r] = tnode[0].value;
        if (p.env_steps) p.env_steps[r] = (int64_t)steps_taken;
        const int rfc = tnode[0].first_child;
        for (int a = 0; a < A; ++a) {
            if (p.root_child_count) p.root_child_count[(long)r * A + a] = rfc >= 0 ? max(tnode[rfc + a].count, 0) : 0;
            if (p.root_child_value) p.root_child_value[(long)r * A + a] = rfc >= 0 ? tnode[rfc + a].value : 0.0;
        }
    }
}

// ---- FOUR ROOTS PER WAVEFRONT, ONE DPP ROW (16 LANES) EACH, ONE MDP PER ROOT (round 6): the batch of highway episodes, each with
// its own finite MDP (trainer/evaluation.py:139-194 x value_iteration.py:29-35).  uct_lone_kernel<.., EACH> gives such a root a
// whole wavefront; but a CDNA SIMD is 16 lanes wide -- a wave64 instruction holds it for 4 cycles whatever it computes -- so
// 4096 lone waves (4 per SIMD) run at a quarter of a lone wave's speed (measured: 0.176 ms, no better than one lane per root).
// Here a root gets exactly the 16 lanes it can use: |A| children scored by |A| lanes of ITS row, 16 rollout actions drawn at a
// time by jump-ahead (lane l of the row: the generator 16 k + l + 1 steps ahead), path nodes backed up one per lane; 4096 roots
// are 1024 wavefronts = one per SIMD, every instruction working for four roots.  Row-uniform values (state, depth, the
// generator) live in vector registers, identical in the 16 lanes of a row; cross-lane traffic is DPP inside a row
// (row_newbcast, quad_perm / row_mirror reductions) and one ds_bpermute where the source lane is data-dependent.
// Per root in LDS: the MDP as {uint16 local next state | bit 15 terminal[next]} + the rewards (f64), the tree, the path.
// Same trees, plans, statistics and generator records as every other variant.
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ double row16_max(double u) // every lane: the maximum over the 16 lanes of its DPP row
{
    u = max_step<0xB1, 0xf>(u);  // quad_perm:[1,0,3,2]
    u = max_step<0x4E, 0xf>(u);  // quad_perm:[2,3,0,1]
    u = max_step<0x141, 0xf>(u); // row_half_mirror
    u = max_step<0x140, 0xf>(u); // row_mirror
    return u;
}
// the maximum over lanes 0 .. |A| - 1 of a row (the lanes that score a level's children; the others hold -inf), in every one of them
template <int AT>
__device__ __forceinline__ double row_children_max(double u)
{
    u = max_step<0xB1, 0xf>(u);                 // quad_perm:[1,0,3,2]
    if (AT > 2) u = max_step<0x4E, 0xf>(u);     // quad_perm:[2,3,0,1]
    if (AT > 4) u = max_step<0x141, 0xf>(u);    // row_half_mirror: lanes 0..7 <-> 7..0
    return u;
}
constexpr int kRowRoots = 4; // roots per wavefront
// SHARED (later in round 6): the same kernel for ONE model shared by every root -- SURVEY 8(d)'s own batch, 4096 roots of the
// headline table.  uct_kernel<.., QD> keeps that model's compact tables in LDS (150 of 160 KB) and so has to leave the trees in
// global memory: a dependent L2 round trip per scored level and per backed-up node, a third of a small plan.  Here the workgroup
// stages ONE copy of the transitions (uint16 next | terminal bit: 2 B per (s, a), 100 KB for S = 10 000) and the sixteen trees
// of its four wavefronts take the place of the reward tables; a step's reward comes from the model's 16-byte records in L2 --
// requested when the state chain reaches the step, added to the return IN ORDER one level later (selection) or after the round
// of sixteen steps (rollout), so its latency is paid once per round, not per step.
template <int AT, bool SHARED = false>
__global__ __launch_bounds__(SHARED ? 1024 : 256) void uct_row_kernel(UctArgs p)
{
    static_assert(AT >= 2 && AT <= 8, "|A| with a compile-time specialisation");
    constexpr int A = AT, NTH = AT - 1;
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    // (a workgroup is blockDim.x / 64 wavefronts sharing one copy of the per-call tables; tools/wave_placement.hip: the dispatcher
    // spreads single-wave workgroups over the SIMDs just as well -- 1024 of them land one per SIMD -- so 1, 2 or 4 waves per
    // workgroup run at the same speed)
    // SHARED: a workgroup of sixteen wavefronts stages the model; p.waves of them stay to plan (four roots each)
    const int tid = threadIdx.x, lane = tid & 63, l16 = tid & 15, nthreads = blockDim.x;
    // SHARED: p.lanes = roots per wavefront, 4 or 2 (two: rows 2 and 3 of every wave idle -- twice the waves for the same roots,
    // two per SIMD, which fill each other's issue gaps; their lanes alias the wave's last live row and write nothing)
    const int rpw = SHARED ? p.lanes : kRowRoots;
    const int nrows = SHARED ? p.waves * rpw : nthreads >> 4;         // roots of this workgroup
    const bool row_used = !SHARED || ((tid >> 4) & 3) < rpw;
    const int row = SHARED ? min((tid >> 6) * rpw + min((tid >> 4) & 3, rpw - 1), nrows - 1) : tid >> 4;
    const int H = p.horizon, E = p.episodes, TE = p.table_n;
    double *gpow = lds_d;                   // [H + 1]  gamma ** h
    double *tp = gpow + (H + 1) + A;        // [A]      temperature * |A| * prior[a]
    double *rcp = tp + A;                   // [TE + 1]  1.0 / n
    double *tpdiv = rcp + (TE + 1);         // [A][TE+2] temperature * |A| * prior[a] / n
    const int ntab = (H + 1) + 2 * A + (TE + 1) + A * (TE + 2);
    const int ntab2 = (ntab + 1) & ~1;
    // (SHARED: no reward tables, one transition table for the workgroup)
    const int SA = (SHARED ? p.S : p.Sb) * A, SA2 = SHARED ? 0 : (SA + 1) & ~1, SA8 = (SA + 7) & ~7, PH = (H + 4) & ~3;
    double *rew = lds_d + ntab2 + row * SA2;                                                    // [rows][SA2] rewards
    UctNode *tnode = reinterpret_cast<UctNode *>(lds_d + ntab2 + nrows * SA2) + row * p.cap;    // [rows][cap] trees
    uint32_t *jump = reinterpret_cast<uint32_t *>(reinterpret_cast<UctNode *>(lds_d + ntab2 + nrows * SA2) + nrows * p.cap); // [H + 5][8]
    int32_t *path = reinterpret_cast<int32_t *>(jump + (H + 5) * 8) + row * PH;                 // [rows][PH] path node ids
    uint16_t *t16 = reinterpret_cast<uint16_t *>(reinterpret_cast<int32_t *>(jump + (H + 5) * 8) + nrows * PH) + (SHARED ? 0 : row * SA8); // [rows][SA8]
    const int r = blockIdx.x * nrows + row;
    const bool live = row_used && r < p.n_roots; // (a last wavefront's spare rows run along on the last root's data and write nothing)
    const int rr = live ? r : p.n_roots - 1;
    const int32_t s0g = p.root_state[rr];     // global state of the batch model
    const int32_t sbase = SHARED ? 0 : (s0g / p.Sb) * p.Sb; // first global state of this root's MDP
    for (int i = tid; i < (H + 5) * 8; i += nthreads) jump[i] = p.jump[i];
    for (int i = tid; i < ntab; i += nthreads) lds_d[i] = p.tab[i];
    if (SHARED) {
        // the model's compact transitions (pack_t16: next | terminal[next] << 15), 16 bytes a lane where the tail allows
        const int n8 = SA >> 3;
        const uint4 *src = reinterpret_cast<const uint4 *>(p.t16);
        uint4 *dst = reinterpret_cast<uint4 *>(t16);
#pragma unroll 4
        for (int i = tid; i < n8; i += nthreads) dst[i] = src[i];
        for (int i = (n8 << 3) + tid; i < SA; i += nthreads) t16[i] = p.t16[i];
    } else {
        const Rec *src = p.rec + (long)sbase * A;
#pragma unroll 4
        for (int i = l16; i < SA; i += 16) {
            const Rec rc = src[i];
            t16[i] = (uint16_t)((uint32_t)(rc.next - sbase) | ((rc.flags & 2u) ? 0x8000u : 0u));
            rew[i] = rc.reward;
        }
    }
    // SHARED: reward[s, a] from the 16-byte record in global memory (L2-resident: 16 S |A| bytes)
    auto reward_of = [&](unsigned idx) -> double { return SHARED ? p.rec[idx].reward : rew[idx]; };
    __syncthreads();
    if (SHARED && tid >= p.waves * 64) return;          // (the staging wavefronts)
    auto explore = [&](int a, int cnt1) { return cnt1 <= TE + 1 ? tpdiv[a * (TE + 2) + cnt1] : tp[a] / (double)cnt1; };
    auto inv = [&](int c) { return c <= TE ? rcp[c] : 1.0 / (double)c; };
    auto lds_sync = [] { // this wave's LDS writes before its later reads, for the compiler (the hardware keeps a wave's LDS order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    Pcg64 g;                                  // (the 16 lanes of a row hold the same generator)
    g.load(p.rng + (long)rr * 6);
    const int32_t s0 = s0g - sbase;
    const int32_t st0 = p.root_steps ? p.root_steps[rr] : 0;
    const bool root_term = (p.rec[(long)s0g * A].flags & 1u) != 0; // terminal flag of the root state itself ("source" rule)
    int n_nodes = 1, steps_taken = 0;
    if (l16 == 0 && row_used) { UctNode n; n.value = 0.0; n.count = 0; n.first_child = -1; tnode[0] = n; path[0] = 0; } // mcts.py:129-130 reset()
    lds_sync();
    const int la = l16 < A ? l16 : 0;
    const int row_lane0 = lane & 48;
    // the first round of a rollout's draws (lane l: the generator l + 1 steps ahead = A^(l+1) state + inc G_(l+1)): the limbs of
    // A^(l+1) and the product inc G_(l+1) are the same for every rollout of the plan -- registers, one 128-bit multiply per round
    uint32_t an0[4];
    uint64_t ig0_lo, ig0_hi;
    {
        const int j1 = l16 + 1 < H ? l16 + 1 : H;
        uint32_t gn0[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { an0[i] = jump[j1 * 8 + i]; gn0[i] = jump[j1 * 8 + 4 + i]; }
        Pcg64::mul128v(g.inc_lo, g.inc_hi, gn0, ig0_lo, ig0_hi);
    }
#ifdef MP_PROFILE
    long long t_sel = 0, t_exp = 0, t_draw = 0, t_walk = 0, t_cap = 0, t_bak = 0, n_lvl = 0, n_rounds = 0;
    const long long t_all0 = clock64();
#endif
    for (int ep = 0; ep < E; ++ep) { // mcts.py:179-184
        PROF_T(c0);
        int32_t s = s0, st = st0;
        int node = 0, depth = 0;
        bool terminal = false, cur_term = root_term;
        double total = 0.0;
        int fc = tnode[0].first_child;
        // SHARED: the reward of a scored level arrives from L2 while the next level is scored; it is added before that level's own
        // (same order of additions).  pend_w = gamma ** depth * [a reward is pending], pend_r = that reward.
        double pend_r = 0.0, pend_g = 0.0;
        bool pend = false;
        // ---- selection, mcts.py:143-149: a level's children one per lane of the row
        bool sel = live && depth < H && fc >= 0 && !terminal;
        while (any64(sel)) {
            const UctNode c = tnode[sel ? fc + la : 0];
            double sc = c.value + explore(la, c.count + 1);   // MCTSNode.selection_strategy, mcts.py:275-286
            if (l16 >= A) sc = -INFINITY;
            const double m = row_children_max<AT>(sc);
            const unsigned long long ball = ballot64(l16 < A && sc == m); // Node.random_argmax, abstract.py:296-311
            unsigned t = (unsigned)(ball >> row_lane0) & 0xffffu;
            const int nt = __popc(t);
            if (any64(sel && nt > 1)) {   // (no row of this wavefront has a tie: no draw, the first maximum is the only one)
                int pick = 0;
                if (sel && nt > 1) pick = (int)g.below((uint32_t)nt);
#pragma unroll
                for (int k = 0; k < A - 1; ++k)
                    if (pick > 0) { t &= t - 1; --pick; }
            }
            const int act = t ? __ffs((int)t) - 1 : 0;
            const unsigned idx = __umul24((unsigned)s, (unsigned)A) + (unsigned)act;
            // the chosen child's first_child, the transition, the reward and gamma ** depth: ONE LDS round trip
            const int nfc = tnode[sel ? fc + act : 0].first_child;
            const uint32_t e = t16[idx];
            const double rw = reward_of(idx), gp = gpow[depth];
            if (SHARED) {
                if (pend) total += pend_g * pend_r;       // the previous level's reward (requested a level ago)
                pend = sel; pend_r = rw; pend_g = gp;
            }
            if (sel) {
                if (!SHARED) total += gp * rw;
                const bool next_term = (e & 0x8000u) != 0;
                terminal = p.done_on_next ? next_term : cur_term;
                cur_term = next_term;
                s = (int32_t)(e & 0x7fffu);
                ++st; ++steps_taken;
                node = fc + act;
                ++depth;
                if (l16 == 0) path[depth] = node;
                fc = nfc;
            }
            sel = sel && depth < H && fc >= 0 && !terminal;
#ifdef MP_PROFILE
            ++n_lvl;
#endif
        }
        if (SHARED && pend) total += pend_g * pend_r;     // the last scored level's reward
        PROF_T(c1);
        // ---- expansion, mcts.py:151-154 / 237-246
        if (live && fc < 0 && depth < H && (!terminal || node == 0)) {
            if (l16 == 0) tnode[node].first_child = n_nodes;
            if (l16 < A) {
                UctNode n;
                n.value = 0.0; n.count = 0; n.first_child = -1;
                tnode[n_nodes + l16] = n;
            }
            n_nodes += A;
        }
        PROF_T(c2);
        // ---- rollout, mcts.py:156-157 / 160-177: rounds of 16 draws (one per lane of the row, by jump-ahead) and 16 steps of the walk
        const bool want = live && !terminal && depth < H;
        if (any64(want)) {
            int n_lim = H - depth;
            if (p.max_steps > 0 && p.max_steps - st < n_lim) n_lim = p.max_steps - st;
            if (n_lim < 1) n_lim = 1; // (the first step is unconditional, as in the reference's loop)
            bool alive = want, captured = false;
            int n = 0;
            for (int k16 = 0; any64(alive); k16 += 16) {
                PROF_T(d0);
                Pcg64 q = g;
                uint32_t act_l;
                {
                    if (k16 == 0) {
                        uint64_t p_lo, p_hi;
                        Pcg64::mul128v(g.s_lo, g.s_hi, an0, p_lo, p_hi);
                        const uint64_t lo = p_lo + ig0_lo;
                        q.s_hi = p_hi + ig0_hi + (lo < p_lo ? 1ULL : 0ULL);
                        q.s_lo = lo;
                    } else {
                        const int j1 = k16 + l16 + 1 < H ? k16 + l16 + 1 : H; // (draws beyond the horizon are never used)
                        uint32_t an[4], gn[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) { an[i] = jump[j1 * 8 + i]; gn[i] = jump[j1 * 8 + 4 + i]; }
                        q.jump(an, gn);
                    }
                    const uint64_t u = q.output();     // searchsorted(cdf, u, 'right') on the raw 64-bit output
                    int act = 0;
#pragma unroll
                    for (int a = 0; a < NTH; ++a) act += p.thr_arg[a] <= u ? 1 : 0;
                    act_l = (uint32_t)min(act, p.thr_valid);
                }
                PROF_T(d1);
                // SHARED: the rewards of the round's steps are requested as the state chain reaches them and summed afterwards, in
                // step order (a root's live steps are a prefix of the round: n_before .. n - 1)
                double rwv[16];
                const int n_before = n;
                // The walk in GROUPS OF FOUR steps, two passes each.  Pass 1 is the state chain alone -- index -> LDS read -> mask, four
                // times, SPECULATIVE: sw steps through the model whether or not the row's rollout is still running (a finished row reads
                // valid records nobody uses) -- with the rewards requested on the way.  Pass 2 is what the four steps mean for the rows
                // still rolling (terminal flags, step limit, the committed state and return).  A lone wave issues in order: with the
                // bookkeeping between two steps of the chain (the first form) every step paid for both.
                int32_t sw = s;
                uint32_t e4[4];
                double rw4[4];
#define MP_ROW_CHAIN(i_, j_)                                                                            \
    {                                                                                                   \
        const int a_i = dpp_mov<0x150 + (i_)>((int)act_l);      /* row_newbcast: lane i_ of the row */   \
        const unsigned idx = __umul24((unsigned)sw, (unsigned)A) + (unsigned)a_i;                       \
        e4[j_] = t16[idx];                                                                              \
        if (SHARED) rwv[i_] = reward_of(idx); else rw4[j_] = rew[idx];                                  \
        sw = (int32_t)(e4[j_] & 0x7fffu);                                                               \
    }
// pass 2, for the group as a whole: which of its four steps the row takes follows from four bits -- the terminal flags of the
// states reached -- and the step limit, so the per-step if / select chain (fifteen instructions a step) becomes bit arithmetic:
// stop bit j = "the rollout ends AFTER step j"; k = the steps taken = first stop bit + 1 (or 4).  The return's additions are the
// reference's, in order: a step not taken adds +0.0, which changes no bit of a sum that is never -0.
#define MP_ROW_BOOK4()                                                                                  \
    {                                                                                                   \
        const uint32_t nt = ((e4[0] >> 15) & 1u) | ((e4[1] >> 14) & 2u) | ((e4[2] >> 13) & 4u) | ((e4[3] >> 12) & 8u);  \
        const uint32_t th = p.done_on_next ? nt : (((nt << 1) | (cur_term ? 1u : 0u)) & 15u);           \
        const int left = n_lim - n;                              /* steps still allowed: >= 1 while alive */ \
        const uint32_t lim = left <= 4 ? 1u << ((left - 1) & 3) : 0u;                                   \
        const uint32_t stop = th | lim;                                                                 \
        const int first = stop ? __ffs((int)stop) - 1 : 4;                                              \
        const int k = alive ? min(first + 1, 4) : 0;                                                    \
        if (!SHARED) {                                                                                  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                             \
                const double add = gpow[min(depth + n + j, H)] * rw4[j];                                \
                total += j < k ? add : 0.0;                                                             \
            }                                                                                           \
        }                                                                                               \
        const uint32_t ek = k <= 1 ? e4[0] : (k == 2 ? e4[1] : (k == 3 ? e4[2] : e4[3]));               \
        s = k > 0 ? (int32_t)(ek & 0x7fffu) : s;                                                        \
        cur_term = k > 0 ? (ek & 0x8000u) != 0 : cur_term;                                              \
        n += k;                                                                                         \
        alive = alive && first >= 4;                                                                    \
    }
#define MP_ROW_WALK4(g_)                                                                                \
    MP_ROW_CHAIN(4 * (g_), 0) MP_ROW_CHAIN(4 * (g_) + 1, 1) MP_ROW_CHAIN(4 * (g_) + 2, 2) MP_ROW_CHAIN(4 * (g_) + 3, 3)     \
    MP_ROW_BOOK4()
                int groups = 1;                                   // groups of four this round walked (wave-uniform)
                MP_ROW_WALK4(0)
                if (any64(alive)) {
                    groups = 2;
                    MP_ROW_WALK4(1)
                    if (any64(alive)) {
                        groups = 3;
                        MP_ROW_WALK4(2)
                        if (any64(alive)) { groups = 4; MP_ROW_WALK4(3) }
                    }
                }
#undef MP_ROW_WALK4
#undef MP_ROW_BOOK4
#undef MP_ROW_CHAIN
                if (SHARED) {
                    // the round's return, in step order, over the groups that were walked; a step the row did not take adds +0.0
                    const int taken = n - n_before;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        if (gq < groups) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int i = 4 * gq + j;
                                const double add = gpow[min(depth + n_before + i, H)] * rwv[i];
                                total += i < taken ? add : 0.0;
                            }
                        }
                    }
                }
                PROF_T(d2);
                // a row whose rollout ended in this round: its generator after the n draws it consumed = the state the lane of
                // its last draw jumped to
                const int src = (row_lane0 + ((n - 1) & 15)) << 2;
                const uint32_t x0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)q.s_lo);
                const uint32_t x1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(q.s_lo >> 32));
                const uint32_t x2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)q.s_hi);
                const uint32_t x3 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(q.s_hi >> 32));
                if (want && !alive && !captured) {
                    g.s_lo = ((uint64_t)x1 << 32) | x0;
                    g.s_hi = ((uint64_t)x3 << 32) | x2;
                    captured = true;
                }
#ifdef MP_PROFILE
                { const long long d3 = clock64(); t_draw += d1 - d0; t_walk += d2 - d1; t_cap += d3 - d2; ++n_rounds; }
#endif
            }
            if (want) { st += n; steps_taken += n; }
        }
        PROF_T(c3);
        // ---- backup, mcts.py:248-265: the same return for every node on the path, one node per lane of the row
        lds_sync();
        for (int d0 = 0; any64(live && d0 <= depth); d0 += 16) {
            const int d = d0 + l16;
            if (live && d <= depth) {
                const int nd = path[d];
                UctNode c = tnode[nd];
                c.count += 1;
                c.value += inv(c.count) * (total - c.value);
                tnode[nd].value = c.value;       // (first_child is left alone: the node may just have been expanded)
                tnode[nd].count = c.count;
            }
        }
        lds_sync();
#ifdef MP_PROFILE
        { const long long c4 = clock64(); t_sel += c1 - c0; t_exp += c2 - c1; t_bak += c4 - c3; }
#endif
    }
#ifdef MP_PROFILE
    if (blockIdx.x == 0 && lane == 0)
        printf("uct_row prof wave0: total=%lld select=%lld (levels %lld) expand=%lld draw=%lld walk=%lld capture=%lld (rounds %lld) backup=%lld\n",
               (long long)(clock64() - t_all0), t_sel, n_lvl, t_exp, t_draw, t_walk, t_cap, n_rounds, t_bak);
#endif
    if (!live) return;
    // the tree, to global memory in the group-interleaved layout (export, re-rooting by step_by_subtree)
    const TreeRef<2, AT> tree = tree_of<2, AT>(p.tree, r, p.cap, A);
    for (int i = l16; i < n_nodes; i += 16) tree[i] = tnode[i];
    if (l16 == 0) {
        g.store(p.rng + (long)r * 6);
        // ---- AbstractPlanner.get_plan (abstract.py:143-156) with MCTSNode.selection_rule (mcts.py:212-218): most visited
        // child, ties -> first maximal value among them
        int len = 0;
        int fc = tnode[0].first_child;
        while (fc >= 0) {
            int mc = tnode[fc].count;
            for (int a = 1; a < A; ++a) mc = max(mc, tnode[fc + a].count);
            int best = -1;
            double bv = 0.0;
            for (int a = 0; a < A; ++a) {
                const UctNode c = tnode[fc + a];
                if (c.count == mc && (best < 0 || c.value > bv)) { best = a; bv = c.value; }
            }
            if (p.plans && len < p.max_plan_len) p.plans[(long)r * p.max_plan_len + len] = best;
            ++len;
            fc = tnode[fc + best].first_child;
        }
        if (p.plans)
            for (int i = len; i < p.max_plan_len; ++i) p.plans[(long)r * p.max_plan_len + i] = -1;
        if (p.plan_len) p.plan_len[r] = len;
        if (p.n_nodes_out) p.n_nodes_out[r] = n_nodes;
        if (p.root_value) p.root_value[r] = tnode[0].value;
        if (p.env_steps) p.env_steps[r] = (int64_t)steps_taken;
        const int rfc = tnode[0].first_child;
        for (int a = 0; a < A; ++a) {
            if (p.root_child_count) p.root_child_count[(long)r * A + a] = rfc >= 0 ? max(tnode[rfc + a].count, 0) : 0;
            if (p.root_child_value) p.root_child_value[(long)r * A + a] = rfc >= 0 ? tnode[rfc + a].value : 0.0;
        }
    }
}

// AbstractPlanner.step_by_subtree (abstract.py:195-206), one root per lane: the subtree of the root's child
// `action` is re-numbered breadth-first into the other tree buffer (children stay contiguous).  While a node
// waits in the BFS queue its first_child field holds its OLD id.  A never-expanded root gives size 0 (fresh tree).
// one thread: the visit count at the end of an action path (mp_uct_path_count)
template <int IL>
__global__ void uct_path_count_kernel(UctNode *trees, int root, int cap, int A, const int32_t *actions, int n, int64_t *out)
{
    const TreeRef<IL, 0> tree = tree_of<IL, 0>(trees, root, cap, A);
    int node = 0;
    for (int i = 0; i < n; ++i) {
        const int fc = tree[node].first_child;
        const int a = actions[i];
        if (fc < 0 || a < 0 || a >= A) { out[0] = -1; return; }
        node = fc + a;
    }
    const int c = tree[node].count;
    out[0] = c < 0 ? -1 : c; // (count < 0: the phantom slot of an action a listed policy does not list)
}

template <int IL>
__global__ __launch_bounds__(64) void uct_reroot_kernel(int n_roots, int A, int cap_old, int cap_new,
                                                        const UctNode *__restrict__ old_trees, UctNode *__restrict__ new_trees,
                                                        const int32_t *n_old, const int32_t *__restrict__ actions,
                                                        int32_t *n_new) // (n_old and n_new may be the same array)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_roots) return;
    const TreeRef<IL, 0> o = tree_of<IL, 0>(const_cast<UctNode *>(old_trees), r, cap_old, A); // (generic ids: operator[])
    const TreeRef<IL, 0> n = tree_of<IL, 0>(new_trees, r, cap_new, A);
    const int a = actions[r];
    // `if action in self.root.children` (abstract.py:201): an unlisted action has a phantom slot, not a child
    if (n_old[r] < 1 || o[0].first_child < 0 || a < 0 || a >= A || o[o[0].first_child + a].count < 0) {
        n_new[r] = 0;
        return;
    }
    int head = 0, tail = 1;
    UctNode first;
    first.first_child = o[0].first_child + a;
    n[0] = first;
    while (head < tail) {
        const UctNode src = o[n[head].first_child];
        UctNode out;
        out.value = src.value; out.count = src.count; out.first_child = -1;
        if (src.first_child >= 0) {
            out.first_child = tail;
            for (int c = 0; c < A; ++c) {
                UctNode q;
                q.value = 0.0; q.count = 0; q.first_child = src.first_child + c;
                n[tail + c] = q;
            }
            tail += A;
        }
        n[head] = out;
        ++head;
    }
    n_new[r] = tail;
}

// Roots per wavefront for the global-table variant.  A root's episodes are one long dependency
// chain, so a batch takes as long as its slowest wavefront; measured on MI355X (4096 roots,
// highway table) dense waves are fastest: 64 lanes 0.41 ms, 16 lanes 0.44 ms, 4 lanes 0.70 ms.
static int uct_lanes_per_wave()
{
    if (const char *e = getenv("MP_UCT_LANES")) {
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) return v;
    }
    return 64;
}

// whether the LDS-resident model is used when nothing is forced (MP_UCT_MODEL unset)
// Measured on MI355X (headline table, 33 x 30; profiles/r04_uct_ldsr.md): kernel ms global / LDS-resident at 8 192 roots
// 0.342 / 0.357, 32 768 0.376 / 0.383, 65 536 0.412 / 0.369, 131 072 0.564 / 0.473, 262 144 1.049 / 0.781, 524 288
// 2.050 / 1.545 -- staging 150 KB per workgroup costs ~25 us and a lone wave per SIMD hides the gather anyway, so the
// variant is the default from one wave per SIMD (four per CU) upwards.
static bool uct_ldsr_default(bool forced, long n_roots, int cus)
{
    if (forced) return true;
    return (n_roots + 63) / 64 >= 4L * (cus > 0 ? cus : 256);
}

template <int AT>
static int uct_launch(const UctArgs &a, bool ldsm, size_t lds, hipStream_t st, bool sp, bool listed, bool ldsr = false, bool spill = false,
                      bool quad = false)
{
    const int roots_per_block = a.waves * a.lanes;
    const dim3 grid((unsigned)((a.n_roots + roots_per_block - 1) / roots_per_block)), block((unsigned)a.waves * 64);
    if (spill) {
        if (a.tree_il == 2) {
            if constexpr (AT > 0) hipLaunchKernelGGL((uct_kernel<AT, ENV_TABLE_SPILL, false, false, 2>), grid, block, lds, st, a);
        } else {
            hipLaunchKernelGGL((uct_kernel<AT, ENV_TABLE_SPILL, false, false, 0>), grid, block, lds, st, a);
        }
    } else if (ldsr && quad) {
        if constexpr (AT > 0) {
            if (lds > 64 * 1024)
                MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(uct_kernel<AT, ENV_TABLE_LDSR, false, false, 2, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((uct_kernel<AT, ENV_TABLE_LDSR, false, false, 2, true>), grid, block, lds, st, a);
        }
    } else if (ldsr) {
        if constexpr (AT > 0) {
            if (lds > 64 * 1024)
                MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(uct_kernel<AT, ENV_TABLE_LDSR, false, false, 2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((uct_kernel<AT, ENV_TABLE_LDSR, false, false, 2>), grid, block, lds, st, a);
        }
    } else if (sp && listed) {
        if constexpr (AT > 0) {
            if (a.tree_il == 2) hipLaunchKernelGGL((uct_kernel<AT, ENV_TABLE, true, true, 2>), grid, block, lds, st, a);
            else if (a.tree_il == 1) hipLaunchKernelGGL((uct_kernel<AT, ENV_TABLE, true, true, 1>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((uct_kernel<AT, ENV_TABLE, true, true, 0>), grid, block, lds, st, a);
        }
    } else if (sp) {
        if constexpr (AT > 0) {
            if (a.tree_il == 2) hipLaunchKernelGGL((uct_kernel<AT, ENV_TABLE, true, false, 2>), grid, block, lds, st, a);
            else if (a.tree_il == 1) hipLaunchKernelGGL((uct_kernel<AT, ENV_TABLE, true, false, 1>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((uct_kernel<AT, ENV_TABLE, true, false, 0>), grid, block, lds, st, a);
        }
    } else if (ldsm) {
        if (lds > 64 * 1024)
            MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(uct_kernel<AT, ENV_TABLE_LDS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((uct_kernel<AT, ENV_TABLE_LDS>), grid, block, lds, st, a);
    } else if (a.tree_il == 2) {
        if constexpr (AT > 0) hipLaunchKernelGGL((uct_kernel<AT, ENV_TABLE, false, false, 2>), grid, block, lds, st, a);
    } else if (a.tree_il == 1) {
        hipLaunchKernelGGL((uct_kernel<AT, ENV_TABLE, false, false, 1>), grid, block, lds, st, a);
    } else {
        hipLaunchKernelGGL((uct_kernel<AT, ENV_TABLE>), grid, block, lds, st, a);
    }
    return MP_OK;
}

// ---- per-state policies fused ON THE DEVICE (round 6).  mp_policy_load's host loops (download the model's records, S * |A|
// thresholds and fused records in C++, six hipMalloc'ed arrays uploaded) cost 90 ms per call on the batch model of 4096
// highway episodes (491 520 global states) -- per step of the per-episode evaluation loop, whose tables change at every step.
// The same arithmetic as kernels: one thread per state for the prior row, the sampling thresholds ceil(cdf * 2^53) (numpy's
// cumsum order, IEEE division) and the listed-action mask; one thread per (s, a) for the fused record.  `rows` < S: the
// distributions are given for ONE MDP of a batch model (local states) and apply to every MDP -- the planner's own policies on a
// batch of environments that restrict their actions identically.
struct PolBuild {
    int S, A, stride, frq, rows, shift, can_pack;
    const double *prior, *rollout;   // [rows][A]
    const uint8_t *listed, *slot;    // [rows][A] or nullptr
    const Rec *rec;                  // [S * A]
    double *hp;                      // [S][stride]
    uint64_t *ht;                    // [S][stride]
    uint32_t *lmask;                 // [S]
    uint32_t *hf, *hfr, *hp16;       // [S * A][frq * 4], the same by rollout slot (or nullptr), [S * A][4] (or nullptr)
    int32_t *err;                    // {code, state}: 1 negative / NaN probability, 2 rollout row sums to 0, 3 no listed action,
                                     // 4 rollout slots not a permutation, 5 transition out of range
};
__device__ __forceinline__ void pol_err(const PolBuild &b, int code, int s)
{
    if (atomicCAS(b.err, 0, code) == 0) b.err[1] = s;
}
__global__ void policy_rows_kernel(PolBuild b)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= b.S) return;
    const int A = b.A, r = s % b.rows;
    double cdf[8];
    double acc = 0.0;
    unsigned seen = 0;
    for (int a = 0; a < A; ++a) {
        const int c = b.slot ? (int)b.slot[(size_t)r * A + a] : a;
        if (c >= A || ((seen >> c) & 1u)) { pol_err(b, 4, s); return; }
        seen |= 1u << c;
        const double q = b.rollout[(size_t)r * A + c], pr = b.prior[(size_t)r * A + a];
        if (!(q >= 0.0) || !(pr >= 0.0)) pol_err(b, 1, s);
        b.hp[(size_t)s * b.stride + a] = pr;
        acc += q; cdf[a] = acc;                                                    // numpy cumsum, in the rollout policy's order
    }
    if (!(acc > 0.0)) pol_err(b, 2, s);
    for (int a = 0; a < A; ++a) {
        const double scaled = ceil((cdf[a] / acc) * 9007199254740992.0);           // cdf /= cdf[-1]; ceil(ldexp(., 53))
        b.ht[(size_t)s * b.stride + a] = scaled >= 18446744073709551615.0 ? ~0ULL : (uint64_t)scaled;
    }
    for (int a = A; a < b.stride; ++a) { b.hp[(size_t)s * b.stride + a] = 0.0; b.ht[(size_t)s * b.stride + a] = ~0ULL; }
    uint32_t m = (1u << A) - 1u;
    if (b.listed) {
        m = 0;
        for (int a = 0; a < A; ++a) m |= (b.listed[(size_t)r * A + a] ? 1u : 0u) << a;
        if (!m) pol_err(b, 3, s);
    }
    b.lmask[s] = m;
}
__global__ void policy_frec_kernel(PolBuild b)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)b.S * b.A) return;
    const int A = b.A;
    uint32_t *f = b.hf + (size_t)i * b.frq * 4;
    const uint4 rc = reinterpret_cast<const uint4 *>(b.rec)[i];
    const int nx = (int)rc.x;
    if (nx < 0 || nx >= b.S) { pol_err(b, 5, (int)(i / A)); return; }
    f[0] = rc.x; f[2] = rc.z; f[3] = rc.w;
    f[1] = (rc.y & 0xffu) | (b.lmask[nx] << 8) | (b.lmask[i / A] << 16);
    for (int k = 4; k < b.frq * 4; ++k) f[k] = 0xffffffffu;
    uint32_t th[4] = {1023u, 1023u, 1023u, 1023u};
    for (int a = 0; a + 1 < A; ++a) {
        const uint64_t t53 = b.ht[(size_t)nx * b.stride + a];
        const uint64_t hi = t53 >> b.shift;                                        // top bits of the 53, saturated
        f[4 + a] = hi > 0xffffffffULL ? 0xffffffffu : (uint32_t)hi;
        if (b.can_pack && a < 4) th[a] = (t53 >> 43) > 1023ULL ? 1023u : (uint32_t)(t53 >> 43);
    }
    if (b.can_pack) {
        uint32_t *g = b.hp16 + (size_t)i * 4;
        g[0] = (uint32_t)nx | (th[0] << 20);
        g[1] = (rc.y & 3u) | (th[1] << 2) | (th[2] << 12) | (th[3] << 22);
        g[2] = rc.z; g[3] = rc.w;
    }
}
__global__ void policy_froll_kernel(PolBuild b) // entry (s, k) = the record of (s, column of slot k)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)b.S * b.A) return;
    const int s = (int)(i / b.A), k = (int)(i % b.A), r = s % b.rows;
    const int c = b.slot[(size_t)r * b.A + k];
    if (c >= b.A) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(b.hf) + ((size_t)s * b.A + c) * b.frq;
    uint4 *dst = reinterpret_cast<uint4 *>(b.hfr) + (size_t)i * b.frq;
    for (int q = 0; q < b.frq; ++q) dst[q] = src[q];
}

} // namespace mp

using namespace mp;

// Apply the re-rooting armed by mp_uct_step_tree: every kept tree -> the subtree under its root's child actions[i], written
// into the other tree workspace with node stride cap_new (>= the current tree sizes); sizes are updated in place.
static int uct_reroot_now(mp_ctx *ctx, long cap_new)
{
    const int n_roots = ctx->tree.n_roots, A = ctx->tree.A;
    const int old_slot = ctx->tree.buf ? WS_TREE2 : WS_TREE0, new_slot = ctx->tree.buf ? WS_TREE0 : WS_TREE2;
    UctNode *nw = nullptr;
    MP_TRY(ws_get(ctx, new_slot, (size_t)((n_roots + 63) & ~63) * tree_stride_alloc(ctx->tree.il, cap_new, A), &nw));
    int32_t *sizes = (int32_t *)ctx->ws[WS_TREE1].p;
    const int32_t *acts = (const int32_t *)ctx->ws[WS_TREE3].p;
    const dim3 rgrid((unsigned)((n_roots + 63) / 64));
    const UctNode *od = (const UctNode *)ctx->ws[old_slot].p;
    if (ctx->tree.il == 2)
        hipLaunchKernelGGL(uct_reroot_kernel<2>, rgrid, dim3(64), 0, ctx->stream, n_roots, A, ctx->tree.cap, (int)cap_new, od, nw, sizes, acts, sizes);
    else if (ctx->tree.il == 1)
        hipLaunchKernelGGL(uct_reroot_kernel<1>, rgrid, dim3(64), 0, ctx->stream, n_roots, A, ctx->tree.cap, (int)cap_new, od, nw, sizes, acts, sizes);
    else
        hipLaunchKernelGGL(uct_reroot_kernel<0>, rgrid, dim3(64), 0, ctx->stream, n_roots, A, ctx->tree.cap, (int)cap_new, od, nw, sizes, acts, sizes);
    MP_HIP(hipGetLastError());
    ctx->tree.buf ^= 1;
    ctx->tree.cap = (int)cap_new;
    ctx->tree.armed = false;
    return MP_OK;
}

static int uct_plan_impl(mp_ctx *ctx, mp_model *model, const mp_policy *pol, int32_t n_roots, const void *root_state,
                         const int32_t *root_steps, int32_t episodes, int32_t horizon, double gamma, double temperature,
                         const double *prior_p, const double *rollout_p, uint64_t *rng_state, int32_t max_plan_len,
                         int32_t *plans, int32_t *plan_len, double *root_value, int64_t *root_child_count,
                         double *root_child_value, int64_t *env_steps, int32_t mem)
{
    if (!ctx || !model || !root_state || !rng_state || (!pol && (!prior_p || !rollout_p)))
        return fail(MP_ERR_ARG, "mp_uct_plan: NULL argument");
    const bool cart = model->mode == MP_MODE_CARTPOLE;
    if (pol && !cart && pol->model == model && pol->model_serial == model->serial && pol->ctx == ctx && !pol->frec && pol->A > 8)
        return fail(MP_ERR_ARG, "mp_uct_plan_policy: |A| = %d is not in 2..8: per-state policies over more actions plan through "
                                "mp_uct_plan_stochastic_policy (it takes deterministic tables too)", pol->A);
    if (pol && (cart || pol->model != model || pol->model_serial != model->serial || pol->ctx != ctx || !pol->frec))
        return fail(MP_ERR_ARG, "mp_uct_plan_policy: the policy was not loaded for this model");
    if (model->mode != MP_MODE_DETERMINISTIC && !cart)
        return fail(MP_ERR_MODE, "mp_uct_plan: model mode %d is neither a deterministic table nor CartPole", model->mode);
    if (model->masked && !(pol && pol->listed))
        return fail(MP_ERR_ARG, "mp_uct_plan: the model restricts its action sets (mp_model_set_available): MCTS reads "
                                "availability through its policies -- plan with a policy from mp_policy_load_listed");
    if (n_roots < 1 || episodes < 0 || horizon < 0 || max_plan_len < 0)
        return fail(MP_ERR_ARG, "mp_uct_plan: bad sizes (n_roots=%d episodes=%d horizon=%d)", n_roots, episodes, horizon);
    if (!mem_valid(mem)) return fail(MP_ERR_ARG, "mp_uct_plan: unknown mem flags %d", mem);
    const int A = model->A, H = horizon, E = episodes;
    MP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const long cap = 1 + (long)episodes * A;

    // small per-call tables, computed on the host exactly as Python computes them
    // (the quotient tables live in LDS: 16 KB of them at most; counts beyond take the division in the kernel)
    const int te_max = (int)(16384 / (sizeof(double) * (size_t)(A + 1)));
    const int TE = E < te_max ? E : te_max;
    const size_t ntab = (size_t)(H + 1) + 2 * (size_t)A + (TE + 1) + (size_t)A * (TE + 2);
    std::vector<double> tab(ntab);
    double *gpow = tab.data(), *cdf = gpow + (H + 1), *tpv = cdf + A, *rcp = tpv + A, *tpdiv = rcp + (TE + 1);
    for (int h = 0; h <= H; ++h) gpow[h] = pow(gamma, (double)h);                 // gamma ** h
    double acc = 0.0;
    for (int a = 0; a < A; ++a) { acc += pol ? 1.0 : rollout_p[a]; cdf[a] = acc; } // numpy cumsum
    for (int a = 0; a < A; ++a) cdf[a] /= acc;                                     // cdf /= cdf[-1]
    for (int a = 0; a < A; ++a) {
        // Generator.choice(p=...): idx = searchsorted(cdf, u, 'right') with u = k * 2^-53, k < 2^53 integer;
        // cdf[a] <= u  <=>  ceil(cdf[a] * 2^53) <= k  (the scaling by 2^53 is exact)
        const double scaled = ceil(ldexp(cdf[a], 53));
        const uint64_t t = scaled >= 18446744073709551615.0 ? ~0ULL : (uint64_t)scaled;
        memcpy(&cdf[a], &t, sizeof(t));
    }
    rcp[0] = 0.0;
    for (int n = 1; n <= TE; ++n) rcp[n] = 1.0 / (double)n;                        // mcts.py:255  K / count, K = 1
    for (int a = 0; a < A; ++a) {
        const double tp = temperature * (double)A * (pol ? 0.0 : prior_p[a]);     // mcts.py:286, left to right
        tpv[a] = tp;
        tpdiv[(size_t)a * (TE + 2)] = 0.0;
        for (int n = 1; n <= TE + 1; ++n) tpdiv[(size_t)a * (TE + 2) + n] = tp / (double)n; // ... / (count + 1)
    }
    double *d_tab = nullptr;
    MP_TRY(upload_tables(ctx, 1, tab, &d_tab));

    UctArgs a;
    a.n_roots = n_roots; a.S = model->S; a.A = A; a.episodes = episodes; a.horizon = horizon; a.cap = (int)cap;
    a.table_n = TE; a.rep_shift = 0;
    a.done_on_next = model->done_on_next; a.max_steps = model->max_steps; a.max_plan_len = max_plan_len;
    a.rec = model->rec; a.t16 = model->t16; a.tab = d_tab;
    a.thr_valid = 0;
    for (int i = 0; i < 8; ++i) {
        a.thr_arg[i] = ~0ULL;
        if (i < A) {
            uint64_t t;
            memcpy(&t, &cdf[i], sizeof(uint64_t));
            if (t < (1ULL << 53)) {
                a.thr_arg[i] = t << 11;
                if (i < A - 1) a.thr_valid = i + 1;     // (thresholds are non-decreasing: the valid ones are a prefix)
            }
        }
    }
    a.cp = model->cp; a.cp_sincos = model->cp_sincos; a.root_x = nullptr;
    a.cp_fastdiv = 0; a.cp_inv_tm = 0.0;
    if (cart) {
        // div_by_const's conditions (see the rollout): every parameter a numerator is built from within [2^-50, 2^50], the divisor
        // normal with a significand that is not all ones, the other denominator bounded away from zero.  MP_CART_FASTDIV=0: A/B.
        const mp_cartpole_params &cq = model->cp;
        const double tm = cq.masspole + cq.masscart;
        auto mid = [](double v) { return std::isfinite(v) && fabs(v) >= 0x1p-50 && fabs(v) <= 0x1p50; };
        uint64_t bits;
        memcpy(&bits, &tm, 8);
        const char *e = getenv("MP_CART_FASTDIV");
        if (mid(tm) && tm > 0 && (bits & ((1ULL << 52) - 1)) != (1ULL << 52) - 1 && mid(cq.masspole) && cq.masspole > 0 && mid(cq.length) &&
            cq.length > 0 && mid(cq.force_mag) && mid(cq.gravity) && mid(cq.tau) && cq.length * (4.0 / 3.0 - cq.masspole / tm) >= 0x1p-50 &&
            !(e && atoi(e) == 0)) {
            a.cp_fastdiv = 1;
            a.cp_inv_tm = 1.0 / tm;
        }
    }
    a.pol_prior = pol ? pol->prior : nullptr; a.pol_thr = pol ? pol->thr : nullptr; a.pol_frec = pol ? pol->frec : nullptr;
    a.pol_frec_roll = pol ? (pol->frec_roll ? pol->frec_roll : pol->frec) : nullptr;
    a.pol_stride = pol ? pol->stride : 0;
    // 16-byte records for saturated batches (see mp_policy_load)
    bool use16 = pol && pol->packed && n_roots > 16384;
    if (const char *rf = getenv("MP_UCT_POLICY_RECORD")) use16 = pol && pol->packed && rf[0] == 'p' ? true : (rf[0] == 'f' ? false : use16);
    const bool listed = pol && pol->listed;
    if (listed) use16 = false; // the listed-action masks live in the fused records' flags word
    a.pol_shift = pol ? (use16 ? 43 : pol->shift) : 21;
    a.pol_sat = use16 ? 1023u : 0xffffffffu;
    a.pol_packed = use16 ? 1 : 0;
    if (use16) a.pol_frec = pol->frec16;
    a.TA = temperature * (double)A;
    a.temperature = temperature;

    // variant and geometry
    // Measured on MI355X (highway table, 4096 roots): global-record variant 0.338 ms, LDS-table variant 0.376 ms --
    // the per-step chain is instruction-bound (PCG64's 128-bit multiply), not gather-latency bound, so the
    // single-gather variant is the default; MP_UCT_MODEL=lds selects the LDS-resident transition table.
    const char *force = getenv("MP_UCT_MODEL"); // "global" (default) / "lds"
    const char *lay = getenv("MP_UCT_TREE"); // "rootmajor" / "interleaved" / "group": tree layout (TreeRef)
    const bool at_known = A >= 2 && A <= 8;
    // default: group-interleaved wherever |A| has a compile-time specialisation (262 144 roots: 1.05-1.07 ms against
    // 1.16 ms root-major and 1.09 ms interleaved; 4 096 roots and single roots: no difference)
    const int want_il = !lay ? (at_known ? 2 : 0) : (lay[0] == 'i' ? 1 : (lay[0] == 'g' && at_known ? 2 : 0));
    bool ldsm = !cart && !pol && model->t16 != nullptr && force && !strcmp(force, "lds"); // (the LDS variant keeps root-major trees)
    // LDS-RESIDENT model (transitions + reward indices + reward table in LDS, nothing of an env step in global memory):
    // table models with S < 32768, at most 256 distinct rewards, |A| with a specialisation, whose 3 B per (s,a) fit the
    // CU's LDS next to the per-call tables.  MP_UCT_MODEL=ldsr forces it on, =global off.
    bool ldsr = !cart && !pol && at_known && model->t16 != nullptr && model->r8 != nullptr && want_il == 2 &&
                !(force && strcmp(force, "ldsr") != 0) && uct_ldsr_default(force != nullptr, n_roots, ctx->prop.multiProcessorCount);
    const size_t sa16 = ((size_t)model->S * A + 15) & ~(size_t)15;
    const size_t lds_ldsr = (((ntab + 1) & ~(size_t)1) + (((size_t)model->n_rdict + 1) & ~(size_t)1)) * sizeof(double) +
                            (((size_t)model->S * A + 7) & ~(size_t)7) * 2 + sa16;
    if (ldsr && lds_ldsr > kLdsBytes) {
        if (force && !strcmp(force, "ldsr")) return fail(MP_ERR_ARG, "mp_uct_plan: model does not fit LDS (%zu B)", lds_ldsr);
        ldsr = false;
    }
    if (ldsr) ldsm = false;
    // FOUR LANES PER ROOT (uct_kernel<..., QD>): small batches of an LDS-resident model -- at most one wave per SIMD's worth of
    // roots (16 roots per wave: 16 384 roots on 256 CUs).  MP_UCT_QUAD=1 / 0 forces it on / off.
    bool quad = false;
    const size_t lds_quad = lds_ldsr + (size_t)(H + 5) * 32;   // jump table: n = 0 .. H (and the lanes' 1 .. 4)
    if (!cart && !pol && at_known && model->t16 != nullptr && model->r8 != nullptr && want_il == 2 && H >= 1 && H <= 255 &&
        lds_quad <= kLdsBytes) {
        const long cus_q = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
        const char *qe = getenv("MP_UCT_QUAD");
        // measured (profiles/r05_uct_small_batch.md): 64 .. 16 384 roots 1.2-1.25x faster than the one-lane-per-root gather
        // kernel, a single root level with it (0.186 against 0.183 ms: its chain is selection + backup in global memory)
        quad = qe ? atoi(qe) != 0 : (!force && n_roots >= 16 && ((long)n_roots + 15) / 16 <= 4 * cus_q);
    }
    if (quad) { ldsr = true; ldsm = false; }
    // ONE ROOT PER WORKGROUP (uct_lone_kernel): batches of at most one root per CU -- a single agent's act() above all -- on fresh
    // trees: model, tables AND tree in LDS, the whole wavefront working for the root.  MP_UCT_LONE=1 / 0 forces it on (any batch) / off.
    bool lone = false;
    int lone_w = 0;                 // > 0: the multi-wavefront form (MW), that many planning wavefronts per workgroup
    const size_t lds_lone = lds_quad + (size_t)cap * (sizeof(UctNode) + sizeof(double)) + (size_t)(H + 1) * sizeof(int32_t) + 16;
    auto lds_lone_mw_of = [&](int w) {     // tables + the transitions + w x (jump table + tree + exploration terms)
        return ((ntab + 1) & ~(size_t)1) * sizeof(double) + (((size_t)model->S * A + 7) & ~(size_t)7) * 2 +
               (size_t)w * ((((size_t)(H + 5) * 8 + (size_t)cap * 6 + 3) & ~(size_t)3) * 4) + 16;
    };
    {
        const bool will_continue = ctx->tree.armed && ctx->tree.kind == 1 && ctx->tree.n_roots == n_roots && ctx->tree.A == A;
        if (!cart && !pol && at_known && model->t16 != nullptr && model->r8 != nullptr && want_il == 2 && H >= 1 && H <= 63 &&
            lds_lone <= kLdsBytes && !will_continue) {
            const char *le = getenv("MP_UCT_LONE");
            const long cus_l = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
            // One workgroup per CU for a model that fills its LDS (the headline table: 256 roots 0.088 ms); for smaller models as many
            // as fit a CU's LDS together, up to two planning wavefronts per SIMD (tools/uct_small_batch.py with UCT_SB_SHAPE, round 6:
            // S = 1 000: 1 024 roots 0.068 ms against the row kernel's 0.090; S = 250: 2 048 roots 0.083 / 0.094; S = 3 000: 512
            // roots 0.088 / 0.128 -- and slower than the rows as soon as the workgroups need a second round)
            const long fit_l = (long)(kLdsBytes / lds_lone);
            lone = le ? atoi(le) != 0 : (!force && !getenv("MP_UCT_QUAD") && n_roots <= cus_l * (fit_l < 1 ? 1 : (fit_l > 8 ? 8 : fit_l)));
        }
        // ... and for a model that fills the LDS (the headline table: one such workgroup per CU), SEVERAL planning wavefronts per
        // workgroup around one copy of the transitions, the rewards from the records in L2 (uct_lone_kernel<.., MW>): batches of up
        // to 8 roots per CU (two wavefronts per SIMD) -- and, with ONE planning wavefront, small batches of a model the plain form cannot
        // take.  MP_UCT_LONE_WAVES=1 / 2 / 4 / 8 forces the form (any batch), 0 turns it off.
        const char *we = getenv("MP_UCT_LONE_WAVES");
        const bool we_on = we && (atoi(we) == 1 || atoi(we) == 2 || atoi(we) == 4 || atoi(we) == 8);
        if (!cart && !pol && at_known && model->t16 != nullptr && model->NB <= 1 && want_il == 2 && H >= 1 && H <= 63 && !will_continue &&
            (!lone || we_on) && !(getenv("MP_UCT_LONE") && atoi(getenv("MP_UCT_LONE")) == 0)) {
            const long cus_l = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
            int w = 0;
            if (we) { if (we_on) w = atoi(we); }
            else if (!force && !getenv("MP_UCT_QUAD") && !getenv("MP_UCT_ROWS") && !getenv("MP_UCT_PATH") && n_roots <= 8 * cus_l)
                // (at most one root per CU gets here when the plain form is not to be had: no compact reward index -- more than 256
                // distinct rewards -- or transitions + reward indices beyond the LDS; the transitions alone may still fit)
                w = n_roots <= cus_l ? 1 : (n_roots <= 2 * cus_l ? 2 : (n_roots <= 4 * cus_l ? 4 : 8));
            if (w && lds_lone_mw_of(w) <= kLdsBytes) { lone = true; lone_w = w; }
        }
    }
    // ONE MDP PER ROOT (uct_lone_kernel<.., EACH>): a batch model -- every root's own MDP staged into its workgroup's LDS from the
    // 16-byte records (10 B per (s, a)), whatever the batch size.  MP_UCT_EACH=0 keeps the one-lane-per-root gather kernel.
    bool each = false;
    const int Sb = model->NB > 1 && model->Sb > 0 ? model->Sb : model->S;
    const size_t sa_each = (size_t)Sb * A;
    const size_t lds_each = (((ntab + 1) & ~(size_t)1) + ((sa_each + 1) & ~(size_t)1)) * sizeof(double) + ((sa_each + 7) & ~(size_t)7) * 2 +
                            (size_t)(H + 5) * 32 + (size_t)cap * (sizeof(UctNode) + sizeof(double)) + (size_t)(H + 1) * sizeof(int32_t) + 16;
    {
        const bool will_continue = ctx->tree.armed && ctx->tree.kind == 1 && ctx->tree.n_roots == n_roots && ctx->tree.A == A;
        const char *ee = getenv("MP_UCT_EACH");
        if (!cart && !pol && at_known && model->NB > 1 && model->Sb > 0 && model->Sb < 32768 && want_il == 2 && H >= 1 && H <= 63 &&
            lds_each <= kLdsBytes && !will_continue && !force && !(ee && atoi(ee) == 0))
            each = true;
    }
    // ... and FOUR such roots per wavefront, a DPP row each (uct_row_kernel): the default wherever four MDPs + trees fit the LDS of
    // a workgroup.  MP_UCT_ROW=0 keeps a wavefront per root.
    bool rowk = false;
    // waves per workgroup: four (one per SIMD of the CU) while their MDPs + trees fit the LDS, else two / one
    auto lds_row_of = [&](int rows_wg) {
        return (((ntab + 1) & ~(size_t)1) + rows_wg * ((sa_each + 1) & ~(size_t)1)) * sizeof(double) + rows_wg * (size_t)cap * sizeof(UctNode) +
               (size_t)(H + 5) * 32 + rows_wg * (size_t)((H + 4) & ~3) * sizeof(int32_t) + rows_wg * ((sa_each + 7) & ~(size_t)7) * 2;
    };
    int row_waves = 4;
    if (const char *e = getenv("MP_UCT_ROW_WAVES")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) row_waves = v; }
    while (row_waves > 1 && lds_row_of(kRowRoots * row_waves) > kLdsBytes) row_waves >>= 1;
    const size_t lds_row = lds_row_of(kRowRoots * row_waves);
    {
        const char *re = getenv("MP_UCT_ROW");
        const bool will_continue = ctx->tree.armed && ctx->tree.kind == 1 && ctx->tree.n_roots == n_roots && ctx->tree.A == A;
        if (!cart && !pol && at_known && model->NB > 1 && model->Sb > 0 && model->Sb < 32768 && want_il == 2 && H >= 1 && H <= 255 &&
            lds_row <= kLdsBytes && !will_continue && !force && !(getenv("MP_UCT_EACH") && atoi(getenv("MP_UCT_EACH")) == 0) &&
            !(re && atoi(re) == 0)) {
            // measured (tools/each_ab.sh, S = 120, 33 x 30, after the lone kernel's rebuild in round 6): a wavefront per root against
            // four roots per wavefront -- 256 roots 0.062 ms (lone), 512: 0.065 / 0.091, 1024: 0.067 / 0.090, 2048: 0.077 / 0.090,
            // 4096: 0.143 / 0.098.  Up to two wavefronts per SIMD (and while their workgroups' LDS fits the CU together) a root is
            // served best by a whole wavefront; beyond, the rows win (the SIMD is 16 lanes wide: four lone waves run at a quarter each).
            const long cus_r = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
            const long fit_cu = lds_each > 0 ? (long)(kLdsBytes / lds_each) : 1;
            const long lone_roots = cus_r * (fit_cu < 1 ? 1 : (fit_cu > 8 ? 8 : fit_cu));
            rowk = (re && atoi(re) != 0) || !each || n_roots > lone_roots;
        }
    }
    if (rowk) each = false;
    // ... and the SHARED-model form of the row kernel (uct_row_kernel<.., true>): batches between one root per CU (uct_lone_kernel)
    // and sixteen (one round of four-wave workgroups) of a model whose transitions + the workgroup's trees fit the LDS.
    // MP_UCT_ROWS=1 / 0 forces it on (any batch size) / off.
    bool rowsh = false;
    int rowsh_waves = 4, rowsh_rpw = kRowRoots;   // planning wavefronts per workgroup, roots per wavefront
    size_t lds_rowsh = 0;
    {
        auto lds_rowsh_of = [&](int rows_wg) {
            return ((ntab + 1) & ~(size_t)1) * sizeof(double) + rows_wg * (size_t)cap * sizeof(UctNode) + (size_t)(H + 5) * 32 +
                   rows_wg * (size_t)((H + 4) & ~3) * sizeof(int32_t) + (((size_t)model->S * A + 7) & ~(size_t)7) * 2;
        };
        const long cus_s = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
        // two roots per wavefront (rows 2 and 3 idle) while that still fits one wave per SIMD: a wave's rounds then run to the slower
        // of 2 roots instead of 4 (2048 roots 0.197 -> 0.187 ms, 1024 roots 0.195 -> 0.184); two such waves per SIMD for 4096 roots
        // measure the same as one wave of four (0.201 against 0.197)
        if ((long)n_roots <= 2L * 4 * cus_s) rowsh_rpw = 2;
        // as few waves per workgroup as still put the batch on the chip in one round (fewer roots behind one staged copy)
        while (rowsh_waves > 1 && ((long)n_roots + rowsh_rpw * (rowsh_waves / 2) - 1) / (rowsh_rpw * (rowsh_waves / 2)) <= cus_s) rowsh_waves >>= 1;
        if (const char *e = getenv("MP_UCT_ROW_WAVES")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) rowsh_waves = v; }
        if (const char *e = getenv("MP_UCT_ROW_ROOTS")) { const int v = atoi(e); if (v == 2 || v == 4) rowsh_rpw = v; }
        while (rowsh_waves > 1 && lds_rowsh_of(rowsh_rpw * rowsh_waves) > kLdsBytes) rowsh_waves >>= 1;
        lds_rowsh = lds_rowsh_of(rowsh_rpw * rowsh_waves);
        const char *se = getenv("MP_UCT_ROWS");
        const bool will_continue = ctx->tree.armed && ctx->tree.kind == 1 && ctx->tree.n_roots == n_roots && ctx->tree.A == A;
        if (!cart && !pol && at_known && model->t16 != nullptr && model->NB <= 1 && want_il == 2 && H >= 1 && H <= 255 &&
            lds_rowsh <= kLdsBytes && !will_continue && !force && !rowk && !each)
            rowsh = se ? atoi(se) != 0 : (!lone && !getenv("MP_UCT_QUAD") && !getenv("MP_UCT_PATH") && n_roots >= 16 && n_roots <= 4L * kRowRoots * cus_s);
    }
    if (each || rowk || rowsh) lone = true;
    if (lone) { quad = false; ldsr = false; ldsm = false; }
    a.Sb = Sb;
    a.jump = nullptr;
    if (quad || lone || cart) {
        // limbs of A^n and G_n = 1 + A + ... + A^(n-1) (mod 2^128), n = 0..H: the generator after n draws is A^n state + inc G_n
        if (ctx->jump_entries < H + 5) {
            typedef unsigned __int128 u128;
            const u128 mult = ((u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
            const int n_e = H + 5 > 64 ? H + 5 : 64;
            // (kept in the context: the upload is asynchronous on the planner's stream -- a caller that runs the planners
            // from a stream with pending waits, e.g. ShardedDevicePlan's double-buffered exchange, must not be synchronised here)
            ctx->jump_host.emplace_back((size_t)n_e * 8, 0u);
            std::vector<uint32_t> &tabj = ctx->jump_host.back();
            u128 an = 1, gn = 0;
            for (int n = 0; n < n_e; ++n) {
                for (int i = 0; i < 4; ++i) { tabj[(size_t)n * 8 + i] = (uint32_t)(an >> (32 * i)); tabj[(size_t)n * 8 + 4 + i] = (uint32_t)(gn >> (32 * i)); }
                gn = gn * mult + 1;
                an = an * mult;
            }
            uint32_t *dj = nullptr;
            ctx->jump_entries = 0;
            MP_TRY(ws_get(ctx, WS_JUMP, tabj.size(), &dj));
            MP_HIP(hipMemcpyAsync(dj, tabj.data(), tabj.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            ctx->jump_entries = n_e;
        }
        a.jump = static_cast<const uint32_t *>(ctx->ws[WS_JUMP].p);
    }
    a.r8 = model->r8; a.rdict = model->rdict; a.n_rdict = model->n_rdict; a.path_spill = nullptr; a.spill_stride = 0;
    a.lanes = quad ? 16 : ((ldsm || ldsr) ? 64 : uct_lanes_per_wave());
    // LDS variant: few roots -> 4 waves per workgroup (one per SIMD); big batches -> 16
    a.waves = ldsm ? ((long)n_roots >= 64L * 16 * ctx->prop.multiProcessorCount ? 16 : 4) : 1;
    if (rowsh) { a.waves = rowsh_waves; a.lanes = rowsh_rpw; }   // (the wavefronts of a workgroup that plan; the launch adds staging waves)
    if (cart) {
        // CartPole (round 6): a root is one lane, and a wavefront takes as long as its SLOWEST root's episodes -- rollouts end when
        // the pole falls, at very different lengths -- so a small batch is spread over more wavefronts of FEWER roots: 16 per
        // wave (4096 roots: 256 waves; 0.564 -> 0.550 ms).  Not fewer: wavefronts with 2 .. 8 active lanes run 1.6 - 3x SLOWER per
        // instruction when the whole chip is busy with them (tools/exec_rate.hip: an f64 division chain 180 -> 527 cycles at 4 active
        // lanes x 1024 waves, unchanged at 16 or 64 lanes or on a single wave; the kernel itself 0.56 -> 1.16 ms at 4 roots per
        // wave although each wave then executes a third fewer instructions: profiles/r06_cartpole.md).
        const long cus_c = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
        // Later in round 6: the slow mode is a matter of how many lanes are ACTIVE, not of how many roots a wave serves (exec_rate.hip
        // with strides: <= 8 active lanes anywhere in the wave are slow about every other run, >= 16 never).  So a root is REPLICATED
        // over 2^rep_shift lanes that all compute its plan -- 4 roots x 16 lanes at 4096 roots: 1024 wavefronts, one per SIMD, 669
        // rollout trips per wave instead of 891 -- and the replicas share the generator work (the rollout block of uct_kernel).
        if (!getenv("MP_UCT_LANES")) {
            long per = ((long)n_roots + 4 * cus_c - 1) / (4 * cus_c);
            a.lanes = 1;
            while (a.lanes < per && a.lanes < 64) a.lanes <<= 1;
        }
        a.waves = 4;
        a.rep_shift = 0;
        while ((a.lanes << (a.rep_shift + 1)) <= 64) ++a.rep_shift;
        if (const char *e = getenv("MP_UCT_CART_REP")) { a.rep_shift = atoi(e); while (a.rep_shift > 0 && (a.lanes << a.rep_shift) > 64) --a.rep_shift; }
        if (a.rep_shift < 2) a.rep_shift = 0;      // (the replicated form works on quads)
        if (const char *e = getenv("MP_UCT_CART_WAVES")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) a.waves = v; }
        // (the per-lane path stack is [H + 1][waves * 64] in LDS: fewer waves per workgroup where four would not fit the CU's LDS --
        // horizons beyond ~150; one wave reaches ~600 steps)
        while (a.waves > 1 && ntab * sizeof(double) + (size_t)(H + 1) * a.waves * 64 * sizeof(int32_t) + 8 + (size_t)MP_SINCOS_ENTRIES * sizeof(double) +
                                      (size_t)(H + 5) * 32 > kLdsBytes)
            a.waves >>= 1;
    }
    if (ldsr) {
        // one workgroup per CU shares the tables: as many waves per workgroup as it takes to put the batch on the chip's
        // CUs (a power of two <= 16, so that chunk boundaries -- multiples of 1024 roots -- are workgroup boundaries)
        const long total_waves = ((long)n_roots + a.lanes - 1) / a.lanes, cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
        long w = (total_waves + cus - 1) / cus;
        if (const char *e = getenv("MP_UCT_LDSR_WAVES")) w = atol(e);
        a.waves = 1;
        while (a.waves < w && a.waves < 16) a.waves <<= 1;
    }
    const size_t lds_base = ntab * sizeof(double) + (size_t)(H + 1) * a.waves * 64 * sizeof(int32_t);
    size_t lds = rowsh ? lds_rowsh : rowk ? lds_row : each ? lds_each : lone ? (lone_w ? lds_lone_mw_of(lone_w) : lds_lone) : quad ? lds_quad : (ldsr ? lds_ldsr : lds_base + (ldsm ? (((size_t)model->S * A * 2 + 15) & ~(size_t)15) + 16 : 0));
    if (cart) lds += 8 + (size_t)MP_SINCOS_ENTRIES * sizeof(double) + (size_t)(H + 5) * 32; // the sin / cos table of libm_sincos.hpp behind the path stack, then the jump table
    if (ldsm && lds > kLdsBytes) {
        if (force && force[0] == 'l') return fail(MP_ERR_ARG, "mp_uct_plan: model does not fit LDS (%zu B)", lds);
        ldsm = false;
        a.lanes = uct_lanes_per_wave(); a.waves = 1;
        lds = ntab * sizeof(double) + (size_t)(H + 1) * 64 * sizeof(int32_t);
    }
    // a path stack that does not fit 64 KB of LDS (horizon >~ 180): registers + the global spill array instead
    bool spill = false;
    if (!lone && !ldsm && !ldsr && lds > 64 * 1024 && !(cart && lds <= kLdsBytes)) {   // (CartPole: the launch raises the kernel's LDS limit)
        const char *lay_now = getenv("MP_UCT_TREE");
        if (cart || pol || (lay_now && lay_now[0] == 'i') || ntab * sizeof(double) > 64 * 1024)
            return fail(MP_ERR_ARG, "mp_uct_plan: horizon %d / episodes %d need %zu B of LDS tables (> 64 KiB)", H, E, lds);
        spill = true;
        lds = ntab * sizeof(double);
    }
    if (const char *e = getenv("MP_UCT_PATH")) // "spill": force the register / global path stack (test hook)
        if (e[0] == 's' && !lone && !ldsm && !ldsr && !cart && !pol && want_il != 1) { spill = true; lds = ntab * sizeof(double); }

    // trees: fresh ones, or (step_strategy "subtree") the kept ones re-rooted into the other buffer with room
    // for this plan's expansions
    int32_t *d_nn = nullptr;
    MP_TRY(ws_get(ctx, WS_TREE1, (size_t)n_roots, &d_nn)); // per-root tree sizes (updated in place by re-rooting and planning)
    a.n_nodes_in = nullptr;
    a.n_nodes_out = d_nn;
    long cap_use = cap;
    const bool cont = ctx->tree.armed && ctx->tree.kind == 1 && ctx->tree.n_roots == n_roots && ctx->tree.A == A;
    if (cont) {
        // a kept subtree only holds nodes created by the last `horizon` plans (a node at depth d survives d re-rootings
        // and d <= horizon), so the stride stays O(horizon * episodes * |A|) however long the episode runs
        // (the largest horizon * episodes of the plans since the last fresh tree, should the caller vary them)
        const long now = 1 + (long)horizon * episodes * A;
        if (now > ctx->tree.kept_bound) ctx->tree.kept_bound = now;
        const long bound = ctx->tree.kept_bound;
        cap_use = (ctx->tree.cap < bound ? (long)ctx->tree.cap : bound) + (long)episodes * A;
        MP_TRY(uct_reroot_now(ctx, cap_use));
        a.tree = (UctNode *)ctx->ws[ctx->tree.buf ? WS_TREE2 : WS_TREE0].p;
        a.n_nodes_in = d_nn;
    } else {
        ctx->tree.armed = false;
        ctx->tree.buf = 0;
        ctx->tree.kept_bound = 1 + (long)horizon * episodes * A;
        ctx->tree.il = (!cart && !ldsm) ? want_il : 0;   // (ldsr implies want_il == 2)
        MP_TRY(ws_get(ctx, WS_TREE0, (size_t)((n_roots + 63) & ~63) * tree_stride_alloc(ctx->tree.il, cap_use, A), &a.tree));
    }
    a.cap = (int)cap_use;
    a.tree_il = ctx->tree.il;
    if (ldsr && ctx->tree.il != 2) { // (a kept tree in another layout: the default kernel and its LDS budget)
        ldsr = false; ldsm = false; quad = false;
        a.lanes = uct_lanes_per_wave();
        a.waves = 1;
        lds = ntab * sizeof(double) + (size_t)(H + 1) * 64 * sizeof(int32_t);
        if (lds > 64 * 1024) { spill = true; lds = ntab * sizeof(double); }
    }
    if (spill && ctx->tree.il == 1) return fail(MP_ERR_ARG, "mp_uct_plan: horizon %d needs the spilled path stack, which the interleaved tree layout does not have", H);
    snprintf(ctx->last_variant, sizeof(ctx->last_variant), "%s", cart ? "uct_cartpole" : (pol ? "uct_policy" : (rowsh ? "uct_row_shared" : rowk ? "uct_row_each" : each ? "uct_lone_each" : lone ? (lone_w ? "uct_lone_mw" : "uct_lone") : quad ? "uct_quad" : ldsr ? "uct_ldsr" : (ldsm ? "uct_lds" : (spill ? "uct_global_spill" : "uct_global")))));
    if (ldsr || spill) {
        a.spill_stride = ((long)n_roots + 63) & ~63L;
        MP_TRY(ws_get(ctx, WS_TREE4, (size_t)(H + 1) * (size_t)a.spill_stride, &a.path_spill));
    }
    ctx->tree.kind = 1; ctx->tree.n_roots = n_roots; ctx->tree.A = A; ctx->tree.cap = (int)cap_use;

    // ---- staging.  Device arrays are used in place; host arrays get device twins (no copy yet: the copies are issued
    // per chunk, below).  The generator records follow their own flag (MP_MEM_RNG_DEVICE: an mp_rng).
    int amem = mem_arrays(mem), rmem = mem_rng(mem);
    const bool host_call = amem == MP_MEM_HOST;       // the call synchronises before it returns
    bool all_pinned = false;                          // every host array lies in mp_host_alloc memory
    // Zero-copy: when EVERY array the caller hands over lives in mp_host_alloc memory (pinned and mapped into the
    // device's address space) the kernel reads the root states from and writes the results to the caller's arrays over
    // the bus, and the whole call is one launch + one synchronisation: no copy is issued at all.
    if (host_call) {
        bool all = true;
        auto alias = [&](const void *p, size_t bytes, auto **out) {
            if (!p) return;
            void *d = pinned_alias(ctx, p, bytes);
            if (!d) all = false; else *out = static_cast<std::remove_reference_t<decltype(*out)>>(d);
        };
        const void *z_rs = nullptr; const int32_t *z_st = nullptr; uint64_t *z_rng = rng_state;
        int32_t *z_plans = nullptr, *z_len = nullptr; double *z_val = nullptr, *z_cv = nullptr; int64_t *z_cc = nullptr, *z_es = nullptr;
        alias(root_state, (size_t)n_roots * (cart ? 32 : 4), &z_rs);
        alias(root_steps, (size_t)n_roots * 4, &z_st);
        if (rmem == MP_MEM_HOST) alias(rng_state, (size_t)n_roots * 48, &z_rng);
        alias(plans, (size_t)n_roots * max_plan_len * 4, &z_plans);
        alias(plan_len, (size_t)n_roots * 4, &z_len);
        alias(root_value, (size_t)n_roots * 8, &z_val);
        alias(root_child_count, (size_t)n_roots * A * 8, &z_cc);
        alias(root_child_value, (size_t)n_roots * A * 8, &z_cv);
        alias(env_steps, (size_t)n_roots * 8, &z_es);
        all_pinned = all;
        // Measured (262 144 roots, 13 MB of results): the kernel writing over the bus 1.52 ms; one launch + asynchronous
        // copies into the same pinned arrays 1.48 ms; two saturating chunks back to back on two streams, the first
        // chunk's results travelling under the second chunk's kernel, 1.35-1.37 ms (device-resident: 1.12).  Small batches
        // (4 096 roots: 0.349 ms zero-copy against 0.346 ms device-resident) are dominated by the calls a copy costs.
        long zc_max = 65536;
        if (const char *e = getenv("MP_ZERO_COPY_MAX")) zc_max = atol(e);
        if (all && !getenv("MP_NO_ZERO_COPY") && n_roots <= zc_max) {
            root_state = z_rs; root_steps = z_st; rng_state = z_rng; plans = z_plans; plan_len = z_len; root_value = z_val;
            root_child_count = z_cc; root_child_value = z_cv; env_steps = z_es;
            amem = MP_MEM_DEVICE; rmem = MP_MEM_DEVICE;     // from here on: device arrays (their aliases)
        }
    }
    const bool host = amem == MP_MEM_HOST;
    int32_t *d_rs = nullptr, *d_st = nullptr;
    double *d_rx = nullptr;
    if (!host) {
        if (cart) d_rx = (double *)const_cast<void *>(root_state); else d_rs = (int32_t *)const_cast<void *>(root_state);
        d_st = const_cast<int32_t *>(root_steps);
    } else {
        if (cart) MP_TRY(ws_get(ctx, WS_IO9, (size_t)n_roots * 4, &d_rx)); else MP_TRY(ws_get(ctx, WS_IO0, (size_t)n_roots, &d_rs));
        if (root_steps) MP_TRY(ws_get(ctx, WS_IO1, (size_t)n_roots, &d_st));
    }
    if (rmem == MP_MEM_DEVICE) a.rng = rng_state; else MP_TRY(ws_get(ctx, WS_IO2, (size_t)n_roots * 6, &a.rng));
    a.root_x = d_rx; a.root_state = d_rs; a.root_steps = d_st;
    // (device twins from the workspaces, NOT stage_out_alloc: that helper would hand back the device alias of a pinned
    // array -- the zero-copy form, decided above for the whole call -- and the chunk copies below would copy it onto itself)
    auto twin = [&](int slot, auto *dst, size_t count, auto **dev) -> int {
        if (!host) { *dev = dst; return MP_OK; }
        if (!dst) { *dev = nullptr; return MP_OK; }
        return ws_get(ctx, slot, count, dev);
    };
    MP_TRY(twin(WS_IO3, plans, (size_t)n_roots * max_plan_len, &a.plans));
    MP_TRY(twin(WS_IO4, plan_len, (size_t)n_roots, &a.plan_len));
    MP_TRY(twin(WS_IO5, root_value, (size_t)n_roots, &a.root_value));
    MP_TRY(twin(WS_IO6, root_child_count, (size_t)n_roots * A, &a.root_child_count));
    MP_TRY(twin(WS_IO7, root_child_value, (size_t)n_roots * A, &a.root_child_value));
    MP_TRY(twin(WS_IO8, env_steps, (size_t)n_roots, &a.env_steps));

    // ---- one chunk of roots [r0, r1): copies in, the kernel, copies out, all on stream `s`.  A chunk is the same launch
    // on shifted pointers (r0 is a multiple of 1024: whole wavefront blocks of the interleaved tree layouts and whole
    // workgroups of every variant), so chunked and unchunked calls leave identical trees and results.
    const long tree_off_per_block = ctx->tree.il == 2 ? (cap_use + A) * 64 : (ctx->tree.il == 1 ? cap_use * 64 : cap_use * 64);
    auto run_chunk = [&](int r0, int r1, hipStream_t s) -> int {
        const size_t cnt = (size_t)(r1 - r0);
        if (host) {
            if (cart) MP_HIP(hipMemcpyAsync(d_rx + (size_t)r0 * 4, (const double *)root_state + (size_t)r0 * 4, cnt * 32, hipMemcpyHostToDevice, s));
            else MP_HIP(hipMemcpyAsync(d_rs + r0, (const int32_t *)root_state + r0, cnt * 4, hipMemcpyHostToDevice, s));
            if (root_steps) MP_HIP(hipMemcpyAsync(d_st + r0, root_steps + r0, cnt * 4, hipMemcpyHostToDevice, s));
        }
        if (rmem == MP_MEM_HOST) MP_HIP(hipMemcpyAsync(a.rng + (size_t)r0 * 6, rng_state + (size_t)r0 * 6, cnt * 48, hipMemcpyHostToDevice, s));
        UctArgs c = a;
        c.n_roots = r1 - r0;
        if (c.root_x) c.root_x += (size_t)r0 * 4;
        if (c.root_state) c.root_state += r0;
        if (c.root_steps) c.root_steps += r0;
        c.rng += (size_t)r0 * 6;
        c.tree += (long)(r0 >> 6) * tree_off_per_block;
        if (c.path_spill) c.path_spill += r0;
        if (c.n_nodes_in) c.n_nodes_in += r0;
        c.n_nodes_out += r0;
        if (c.plans) c.plans += (size_t)r0 * max_plan_len;
        if (c.plan_len) c.plan_len += r0;
        if (c.root_value) c.root_value += r0;
        if (c.root_child_count) c.root_child_count += (size_t)r0 * A;
        if (c.root_child_value) c.root_child_value += (size_t)r0 * A;
        if (c.env_steps) c.env_steps += r0;
        if (cart) {
            const int per_block = c.lanes * c.waves;
            const dim3 grid((unsigned)((c.n_roots + per_block - 1) / per_block)), block(64u * c.waves);
            if (lds > 64 * 1024)
                MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(uct_kernel<2, ENV_CARTPOLE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((uct_kernel<2, ENV_CARTPOLE>), grid, block, lds, s, c);
        } else if (rowsh) {
#define MP_ROWS(k)                                                                                                                 \
    case k:                                                                                                                        \
        if (lds > 64 * 1024)                                                                                                       \
            MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(uct_row_kernel<k, true>),                                   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                    \
        hipLaunchKernelGGL((uct_row_kernel<k, true>), dim3((unsigned)((c.n_roots + rowsh_rpw * rowsh_waves - 1) / (rowsh_rpw * rowsh_waves))), \
                           dim3(1024u), lds, s, c);                                                                                \
        break;
            switch (A) { MP_ROWS(2) MP_ROWS(3) MP_ROWS(4) MP_ROWS(5) MP_ROWS(6) MP_ROWS(7) MP_ROWS(8) default: break; }
#undef MP_ROWS
        } else if (rowk) {
#define MP_ROWK(k)                                                                                                                 \
    case k:                                                                                                                        \
        if (lds > 64 * 1024)                                                                                                       \
            MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(uct_row_kernel<k>),                                         \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                    \
        hipLaunchKernelGGL((uct_row_kernel<k>), dim3((unsigned)((c.n_roots + kRowRoots * row_waves - 1) / (kRowRoots * row_waves))),  \
                           dim3(64u * row_waves), lds, s, c);                                                                      \
        break;
            switch (A) { MP_ROWK(2) MP_ROWK(3) MP_ROWK(4) MP_ROWK(5) MP_ROWK(6) MP_ROWK(7) MP_ROWK(8) default: break; }
#undef MP_ROWK
        } else if (each) {
            // one wave per workgroup while a wave stages its MDP in a few trips; four for larger ones
            const unsigned threads = sa_each <= 4096 ? 64u : 256u;
#define MP_LONE(k)                                                                                                                 \
    case k:                                                                                                                        \
        if (lds > 64 * 1024)                                                                                                       \
            MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(uct_lone_kernel<k, true>),                                  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                    \
        hipLaunchKernelGGL((uct_lone_kernel<k, true>), dim3((unsigned)c.n_roots), dim3(threads), lds, s, c);                      \
        break;
            switch (A) { MP_LONE(2) MP_LONE(3) MP_LONE(4) MP_LONE(5) MP_LONE(6) MP_LONE(7) MP_LONE(8) default: break; }
#undef MP_LONE
        } else if (lone && lone_w) {
            c.waves = lone_w;          // the planning wavefronts of a workgroup (its sixteen stage the model first)
#define MP_LONE(k)                                                                                                                 \
    case k:                                                                                                                        \
        MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(uct_lone_kernel<k, false, true>),                               \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                        \
        hipLaunchKernelGGL((uct_lone_kernel<k, false, true>), dim3((unsigned)((c.n_roots + lone_w - 1) / lone_w)), dim3(1024), lds, s, c); \
        break;
            switch (A) { MP_LONE(2) MP_LONE(3) MP_LONE(4) MP_LONE(5) MP_LONE(6) MP_LONE(7) MP_LONE(8) default: break; }
#undef MP_LONE
        } else if (lone) {
#define MP_LONE(k)                                                                                                                 \
    case k:                                                                                                                        \
        MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(uct_lone_kernel<k>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)lds));                                                                                    \
        hipLaunchKernelGGL((uct_lone_kernel<k>), dim3((unsigned)c.n_roots), dim3(1024), lds, s, c);                               \
        break;
            switch (A) { MP_LONE(2) MP_LONE(3) MP_LONE(4) MP_LONE(5) MP_LONE(6) MP_LONE(7) MP_LONE(8) default: break; }
#undef MP_LONE
        } else
        switch (A) {
        case 2: MP_TRY(uct_launch<2>(c, ldsm, lds, s, pol != nullptr, listed, ldsr, spill, quad)); break;
        case 3: MP_TRY(uct_launch<3>(c, ldsm, lds, s, pol != nullptr, listed, ldsr, spill, quad)); break;
        case 4: MP_TRY(uct_launch<4>(c, ldsm, lds, s, pol != nullptr, listed, ldsr, spill, quad)); break;
        case 5: MP_TRY(uct_launch<5>(c, ldsm, lds, s, pol != nullptr, listed, ldsr, spill, quad)); break;
        case 6: MP_TRY(uct_launch<6>(c, ldsm, lds, s, pol != nullptr, listed, ldsr, spill, quad)); break;
        case 7: MP_TRY(uct_launch<7>(c, ldsm, lds, s, pol != nullptr, listed, ldsr, spill, quad)); break;
        case 8: MP_TRY(uct_launch<8>(c, ldsm, lds, s, pol != nullptr, listed, ldsr, spill, quad)); break;
        default:
            if (pol) return fail(MP_ERR_ARG, "mp_uct_plan_policy: |A| = %d is not in 2..8: per-state policies over more actions plan through "
                                             "mp_uct_plan_stochastic_policy (it takes deterministic tables too)", A);
            MP_TRY(uct_launch<0>(c, ldsm, lds, s, false, false, false, spill));
            break;
        }
        if (rmem == MP_MEM_HOST) MP_HIP(hipMemcpyAsync(rng_state + (size_t)r0 * 6, c.rng, cnt * 48, hipMemcpyDeviceToHost, s));
        if (host) {
            if (plans && max_plan_len) MP_HIP(hipMemcpyAsync(plans + (size_t)r0 * max_plan_len, c.plans, cnt * max_plan_len * 4, hipMemcpyDeviceToHost, s));
            if (plan_len) MP_HIP(hipMemcpyAsync(plan_len + r0, c.plan_len, cnt * 4, hipMemcpyDeviceToHost, s));
            if (root_value) MP_HIP(hipMemcpyAsync(root_value + r0, c.root_value, cnt * 8, hipMemcpyDeviceToHost, s));
            if (root_child_count) MP_HIP(hipMemcpyAsync(root_child_count + (size_t)r0 * A, c.root_child_count, cnt * A * 8, hipMemcpyDeviceToHost, s));
            if (root_child_value) MP_HIP(hipMemcpyAsync(root_child_value + (size_t)r0 * A, c.root_child_value, cnt * A * 8, hipMemcpyDeviceToHost, s));
            if (env_steps) MP_HIP(hipMemcpyAsync(env_steps + r0, c.env_steps, cnt * 8, hipMemcpyDeviceToHost, s));
        }
        return MP_OK;
    };

    // ---- host arrays and a big batch: chunks pipelined over side streams (H2D of chunk i+1 and D2H of chunk i-1 run
    // under the kernel of chunk i, and kernels of different chunks share the chip).  Everything else: one chunk on the
    // ctx stream, as ever.
    // pageable arrays, measured at 262 144 roots: 65 536 x 4 streams 4.1 ms, 32 768 x 4 7.3 ms, 8 192 x 8 12.6 ms (the
    // runtime stages every pageable copy); pinned arrays: two chunks that each fill the chip, see above
    int chunk = 65536, n_streams = 4;
    if (all_pinned) { chunk = (((n_roots + 1) / 2) + 1023) & ~1023; n_streams = 2; }
    if (const char *e = getenv("MP_PIPE_CHUNK")) chunk = atoi(e) > 0 ? ((atoi(e) + 1023) & ~1023) : 0;
    if (const char *e = getenv("MP_PIPE_STREAMS")) { const int v = atoi(e); if (v >= 1 && v <= 8) n_streams = v; }
    const bool piped = host && chunk > 0 && n_roots > chunk;
    MP_TRY(kernels_begin(ctx));
    int launches = 1;
    if (piped) {
        const int n_chunks = (n_roots + chunk - 1) / chunk;
        if (n_streams > n_chunks) n_streams = n_chunks;
        MP_TRY(pipe_fork(ctx, n_streams));
        for (int c = 0; c < n_chunks; ++c) {
            const int r0 = c * chunk, r1 = r0 + chunk < n_roots ? r0 + chunk : n_roots;
            MP_TRY(run_chunk(r0, r1, ctx->pipe[c % n_streams]));
        }
        MP_TRY(pipe_join(ctx, n_streams));
        launches = n_chunks;
    } else {
        MP_TRY(run_chunk(0, n_roots, st));
    }
    MP_TRY(kernels_end(ctx, launches));
    MP_HIP(hipGetLastError());
    if (host_call) MP_HIP(hipStreamSynchronize(st));
    return MP_OK;
}

extern "C" {

int mp_uct_plan(mp_ctx *ctx, mp_model *model, int32_t n_roots, const void *root_state, const int32_t *root_steps,
                int32_t episodes, int32_t horizon, double gamma, double temperature, const double *prior_p,
                const double *rollout_p, uint64_t *rng_state, int32_t max_plan_len, int32_t *plans,
                int32_t *plan_len, double *root_value, int64_t *root_child_count, double *root_child_value,
                int64_t *env_steps, int32_t mem)
{
    return uct_plan_impl(ctx, model, nullptr, n_roots, root_state, root_steps, episodes, horizon, gamma, temperature,
                         prior_p, rollout_p, rng_state, max_plan_len, plans, plan_len, root_value, root_child_count,
                         root_child_value, env_steps, mem);
}

int mp_uct_plan_models(mp_ctx *ctx, mp_model *model, int32_t n_roots, const int32_t *model_index, const int32_t *root_state,
                       const int32_t *root_steps, int32_t episodes, int32_t horizon, double gamma, double temperature,
                       const double *prior_p, const double *rollout_p, uint64_t *rng_state, int32_t max_plan_len, int32_t *plans,
                       int32_t *plan_len, double *root_value, int64_t *root_child_count, double *root_child_value,
                       int64_t *env_steps, int32_t mem)
{
    if (!mem_valid(mem)) return fail(MP_ERR_ARG, "mp_uct_plan_models: unknown mem flags %d", mem);
    std::vector<int32_t> tmp;
    const int32_t *global = nullptr;
    MP_TRY(globalize_roots_arg(ctx, model, n_roots, model_index, root_state, mem, tmp, &global));
    return uct_plan_impl(ctx, model, nullptr, n_roots, global, root_steps, episodes, horizon, gamma, temperature, prior_p, rollout_p,
                         rng_state, max_plan_len, plans, plan_len, root_value, root_child_count, root_child_value, env_steps, mem);
}

int mp_uct_plan_policy(mp_ctx *ctx, mp_model *model, mp_policy *policy, int32_t n_roots, const void *root_state,
                       const int32_t *root_steps, int32_t episodes, int32_t horizon, double gamma, double temperature,
                       uint64_t *rng_state, int32_t max_plan_len, int32_t *plans, int32_t *plan_len, double *root_value,
                       int64_t *root_child_count, double *root_child_value, int64_t *env_steps, int32_t mem)
{
    if (!policy) return fail(MP_ERR_ARG, "mp_uct_plan_policy: policy is NULL");
    return uct_plan_impl(ctx, model, policy, n_roots, root_state, root_steps, episodes, horizon, gamma, temperature,
                         nullptr, nullptr, rng_state, max_plan_len, plans, plan_len, root_value, root_child_count,
                         root_child_value, env_steps, mem);
}

// mcts_with_prior.py:47-62 as tables: per state, the prior row and the integer sampling thresholds of the rollout row
// (computed exactly as mp_uct_plan computes them for one distribution), plus one fused record per (s, a).
int mp_policy_load(mp_ctx *ctx, mp_model *model, const double *prior, const double *rollout, mp_policy **out)
{
    return mp_policy_load_listed(ctx, model, prior, rollout, nullptr, out);
}

int mp_policy_load_listed(mp_ctx *ctx, mp_model *model, const double *prior, const double *rollout, const uint8_t *listed,
                          mp_policy **out)
{
    return mp_policy_load_ordered(ctx, model, prior, rollout, listed, nullptr, out);
}

// rollout_slot (uint8 [S][A], nullptr = identity): the order in which the ROLLOUT policy lists the actions of a state --
// slot k of state s is column rollout_slot[s][k] (a permutation of the columns; zero-probability columns last).  The
// rollout's inverse CDF runs over slots, so the tree (columns: the PRIOR policy's listing order, which decides child
// order and tie-breaks) and the rollout may list the actions differently: policy type `random` lists np.arange(n)
// whatever the environment lists (mcts.py:46-57).  A rollout step never touches the tree -- it needs the next state,
// the reward, the flags and the next state's thresholds -- so its fused records are simply laid out by slot.
int mp_policy_load_ordered(mp_ctx *ctx, mp_model *model, const double *prior, const double *rollout, const uint8_t *listed,
                           const uint8_t *rollout_slot, mp_policy **out)
{
    return mp_policy_load_rows(ctx, model, prior, rollout, listed, rollout_slot, model ? model->S : 0, out);
}

// The fused policy built by kernels (see PolBuild).  One block from the ctx's block cache holds every array.
static int policy_build_device(mp_ctx *ctx, mp_model *model, const double *prior, const double *rollout, const uint8_t *listed,
                               const uint8_t *rollout_slot, int rows, mp_policy **out)
{
    const int S = model->S, A = model->A;
    const int stride = (A + 1) & ~1, frq = 1 + (A - 1 + 3) / 4;
    int shift = 21;
    if (const char *e = getenv("MP_UCT_COARSE_BITS")) {
        const int n = atoi(e);
        if (n >= 1 && n <= 32) shift = 53 - n;
    }
    const bool can_pack = A <= 5 && S <= (1 << 20) && !getenv("MP_UCT_COARSE_BITS") && !listed && !rollout_slot;
    hipStream_t st = ctx->stream;
    // inputs -> workspace
    double *d_prior = nullptr, *d_roll = nullptr;
    uint8_t *d_listed = nullptr, *d_slot = nullptr;
    int32_t *d_err = nullptr;
    const size_t nin = (size_t)rows * A;
    MP_TRY(ws_get(ctx, WS_POL0, nin, &d_prior));
    MP_TRY(ws_get(ctx, WS_POL1, nin, &d_roll));
    MP_TRY(ws_get(ctx, WS_POL4, 2, &d_err));
    MP_HIP(hipMemcpyAsync(d_prior, prior, nin * 8, hipMemcpyHostToDevice, st));
    if (rollout == prior) d_roll = d_prior;
    else MP_HIP(hipMemcpyAsync(d_roll, rollout, nin * 8, hipMemcpyHostToDevice, st));
    if (listed) {
        MP_TRY(ws_get(ctx, WS_POL2, nin, &d_listed));
        MP_HIP(hipMemcpyAsync(d_listed, listed, nin, hipMemcpyHostToDevice, st));
    }
    if (rollout_slot) {
        MP_TRY(ws_get(ctx, WS_POL3, nin, &d_slot));
        MP_HIP(hipMemcpyAsync(d_slot, rollout_slot, nin, hipMemcpyHostToDevice, st));
    }
    MP_HIP(hipMemsetAsync(d_err, 0, 8, st));
    // outputs: one block
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_hp = al((size_t)S * stride * 8), b_ht = b_hp, b_hf = al((size_t)S * A * frq * 16), b_lm = al((size_t)S * 4),
                 b_hfr = rollout_slot ? b_hf : 0, b_p16 = can_pack ? al((size_t)S * A * 16) : 0, b_rs = rollout_slot ? al((size_t)S * A) : 0;
    mp_policy *pol = new (std::nothrow) mp_policy;
    if (!pol) return fail(MP_ERR_ALLOC, "mp_policy_load: out of memory");
    pol->ctx = ctx; pol->model = model; pol->model_serial = model->serial; pol->S = S; pol->A = A; pol->stride = stride; pol->frq = frq;
    pol->shift = shift; pol->packed = can_pack ? 1 : 0; pol->listed = listed ? 1 : 0;
    size_t got = 0;
    if (ctx_block_alloc(ctx, &pol->block, b_hp + b_ht + b_hf + b_lm + b_hfr + b_p16 + b_rs, &got) != hipSuccess) {
        (void)hipGetLastError();
        delete pol;
        return fail(MP_ERR_ALLOC, "mp_policy_load: device allocation failed");
    }
    pol->block_bytes = got;
    char *base = static_cast<char *>(pol->block);
    pol->prior = reinterpret_cast<double *>(base); base += b_hp;
    pol->thr = reinterpret_cast<uint64_t *>(base); base += b_ht;
    pol->frec = reinterpret_cast<uint4 *>(base); base += b_hf;
    pol->lmask = reinterpret_cast<uint32_t *>(base); base += b_lm;
    if (rollout_slot) { pol->frec_roll = reinterpret_cast<uint4 *>(base); base += b_hfr; }
    if (can_pack) { pol->frec16 = reinterpret_cast<uint4 *>(base); base += b_p16; }
    if (rollout_slot) { pol->rslot = reinterpret_cast<uint8_t *>(base); base += b_rs; }
    PolBuild b;
    b.S = S; b.A = A; b.stride = stride; b.frq = frq; b.rows = rows; b.shift = shift; b.can_pack = can_pack ? 1 : 0;
    b.prior = d_prior; b.rollout = d_roll; b.listed = d_listed; b.slot = d_slot; b.rec = model->rec;
    b.hp = pol->prior; b.ht = pol->thr; b.lmask = pol->lmask; b.hf = reinterpret_cast<uint32_t *>(pol->frec);
    b.hfr = reinterpret_cast<uint32_t *>(pol->frec_roll); b.hp16 = reinterpret_cast<uint32_t *>(pol->frec16); b.err = d_err;
    hipLaunchKernelGGL(policy_rows_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, b);
    hipLaunchKernelGGL(policy_frec_kernel, dim3((unsigned)(((size_t)S * A + 255) / 256)), dim3(256), 0, st, b);
    if (rollout_slot) {
        hipLaunchKernelGGL(policy_froll_kernel, dim3((unsigned)(((size_t)S * A + 255) / 256)), dim3(256), 0, st, b);
        // the slots' columns by GLOBAL state (uct_stoch.hip reads them): tile the rows
        for (int off = 0; off < S; off += rows)
            MP_HIP(hipMemcpyAsync(pol->rslot + (size_t)off * A, d_slot, (size_t)(S - off < rows ? S - off : rows) * A, hipMemcpyDeviceToDevice, st));
    }
    int32_t err[2] = {0, 0};
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(err, d_err, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        mp_policy_free(pol);
        return fail(MP_ERR_HIP, "mp_policy_load: %s", hipGetErrorString(e));
    }
    if (err[0]) {
        mp_policy_free(pol);
        const int s_bad = err[1];
        switch (err[0]) {
        case 1: return fail(MP_ERR_ARG, "mp_policy_load: negative or NaN probability in state %d", s_bad);
        case 2: return fail(MP_ERR_ARG, "mp_policy_load: rollout distribution of state %d sums to 0", s_bad);
        case 3: return fail(MP_ERR_ARG, "mp_policy_load_listed: the prior policy lists no action in state %d", s_bad);
        case 4: return fail(MP_ERR_ARG, "mp_policy_load_ordered: rollout_slot of state %d is not a permutation", s_bad);
        default: return fail(MP_ERR_ARG, "mp_policy_load: transition out of range");
        }
    }
    *out = pol;
    return MP_OK;
}

// `rows` = S: one distribution row per state of the model; `rows` = the states of ONE MDP of a batch model
// (mp_model_load_table_batch): the rows are given for local states and serve every MDP of the batch.
int mp_policy_load_rows(mp_ctx *ctx, mp_model *model, const double *prior, const double *rollout, const uint8_t *listed,
                        const uint8_t *rollout_slot, int32_t rows, mp_policy **out)
{
    if (!ctx || !model || !prior || !rollout || !out) return fail(MP_ERR_ARG, "mp_policy_load: NULL argument");
    if (rows != model->S && !(model->NB > 1 && model->Sb > 0 && rows == model->Sb))
        return fail(MP_ERR_ARG, "mp_policy_load_rows: %d rows for a model of %d states (%d MDPs of %d)", rows, model->S, model->NB, model->Sb);
    {
        const bool det = model->mode == MP_MODE_DETERMINISTIC && model->rec && model->A >= 2 && model->A <= 8;
        const char *how = getenv("MP_POLICY_BUILD"); // "host" / "device": test knob
        const bool big = (size_t)model->S * model->A >= 16384;
        if (det && (rows != model->S || (how ? how[0] == 'd' : big))) {
            MP_HIP(hipSetDevice(ctx->device));
            return policy_build_device(ctx, model, prior, rollout, listed, rollout_slot, rows, out);
        }
        if (rows != model->S) return fail(MP_ERR_ARG, "mp_policy_load_rows: tiled rows need a deterministic table model with 2..8 actions");
    }
    const bool stoch = model->mode == MP_MODE_STOCHASTIC || model->mode == MP_MODE_SPARSE; // policies for uct_stoch.hip
    if (!stoch && (model->mode != MP_MODE_DETERMINISTIC || !model->rec))
        return fail(MP_ERR_MODE, "mp_policy_load: per-state policies need a finite-MDP model");
    const int S = model->S, A = model->A;
    if (A < 2) return fail(MP_ERR_ARG, "mp_policy_load: |A| = %d", A);
    if (rollout_slot && A > 256) return fail(MP_ERR_ARG, "mp_policy_load_ordered: rollout slots hold a byte per column (|A| = %d > 256)", A);
    // More than 8 actions (round 4): the policy of the loop-form kernel (uct_stoch.hip plans on deterministic tables too): priors,
    // thresholds, listed actions -- no fused records, whatever the model's mode
    const bool loop_form = stoch || A > 8;
    MP_HIP(hipSetDevice(ctx->device));
    const int stride = (A + 1) & ~1, frq = 1 + (A - 1 + 3) / 4;
    std::vector<double> hp((size_t)S * stride, 0.0);
    std::vector<uint64_t> ht((size_t)S * stride, ~0ULL);
    // column of slot k in state s
    auto col = [&](int s, int k) { return rollout_slot ? (int)rollout_slot[(size_t)s * A + k] : k; };
    if (rollout_slot)
        for (int s = 0; s < S; ++s) {
            std::vector<uint8_t> seen((size_t)A, 0);
            for (int k = 0; k < A; ++k) {
                const int c = rollout_slot[(size_t)s * A + k];
                if (c >= A || seen[(size_t)c]) return fail(MP_ERR_ARG, "mp_policy_load_ordered: rollout_slot of state %d is not a permutation", s);
                seen[(size_t)c] = 1;
            }
        }
    for (int s = 0; s < S; ++s) {
        std::vector<double> cdf((size_t)A);
        double acc = 0.0;
        for (int a = 0; a < A; ++a) {
            const double q = rollout[(size_t)s * A + col(s, a)], pr = prior[(size_t)s * A + a];
            if (!(q >= 0.0) || !(pr >= 0.0)) return fail(MP_ERR_ARG, "mp_policy_load: negative or NaN probability in state %d", s);
            hp[(size_t)s * stride + a] = pr;
            acc += q; cdf[a] = acc;                                                // numpy cumsum, in the rollout policy's order
        }
        if (!(acc > 0.0)) return fail(MP_ERR_ARG, "mp_policy_load: rollout distribution of state %d sums to 0", s);
        for (int a = 0; a < A; ++a) {
            const double scaled = ceil(ldexp(cdf[a] / acc, 53));                   // cdf /= cdf[-1]; see mp_uct_plan
            ht[(size_t)s * stride + a] = scaled >= 18446744073709551615.0 ? ~0ULL : (uint64_t)scaled;
        }
    }
    if (loop_form) {
        // a stochastic model's kernel stores priors in the tree and reads thresholds by state: prior / thresholds / listed
        // masks (+ the rollout slots' columns) are all it needs -- no fused records
        std::vector<uint32_t> lm((size_t)S, A >= 32 ? 0xffffffffu : (1u << A) - 1u);
        std::vector<uint8_t> l8(A > 32 ? (size_t)S * A : 0, 1); // more than 32 actions: a byte per action
        if (listed)
            for (int s = 0; s < S; ++s) {
                uint32_t m = 0;
                bool any = false;
                for (int a = 0; a < A; ++a) {
                    const bool on = listed[(size_t)s * A + a] != 0;
                    any |= on;
                    if (a < 32) m |= (on ? 1u : 0u) << a;
                    if (A > 32) l8[(size_t)s * A + a] = on ? 1 : 0;
                }
                if (!any) return fail(MP_ERR_ARG, "mp_policy_load_listed: the prior policy lists no action in state %d", s);
                lm[(size_t)s] = m;
            }
        mp_policy *pol = new (std::nothrow) mp_policy;
        if (!pol) return fail(MP_ERR_ALLOC, "mp_policy_load: out of memory");
        pol->ctx = ctx; pol->model = model; pol->model_serial = model->serial; pol->S = S; pol->A = A; pol->stride = stride;
        pol->listed = listed ? 1 : 0;
        if (hipMalloc(&pol->prior, hp.size() * 8) != hipSuccess || hipMalloc(&pol->thr, ht.size() * 8) != hipSuccess ||
            hipMalloc(&pol->lmask, lm.size() * 4) != hipSuccess ||
            (!l8.empty() && hipMalloc(&pol->listed8, l8.size()) != hipSuccess) ||
            (rollout_slot && hipMalloc(&pol->rslot, (size_t)S * A) != hipSuccess)) {
            mp_policy_free(pol);
            return fail(MP_ERR_ALLOC, "mp_policy_load: device allocation failed");
        }
        MP_HIP(hipMemcpy(pol->prior, hp.data(), hp.size() * 8, hipMemcpyHostToDevice));
        MP_HIP(hipMemcpy(pol->thr, ht.data(), ht.size() * 8, hipMemcpyHostToDevice));
        MP_HIP(hipMemcpy(pol->lmask, lm.data(), lm.size() * 4, hipMemcpyHostToDevice));
        if (!l8.empty()) MP_HIP(hipMemcpy(pol->listed8, l8.data(), l8.size(), hipMemcpyHostToDevice));
        if (rollout_slot) MP_HIP(hipMemcpy(pol->rslot, rollout_slot, (size_t)S * A, hipMemcpyHostToDevice));
        *out = pol;
        return MP_OK;
    }
    std::vector<Rec> hrec((size_t)S * A);
    MP_HIP(hipStreamSynchronize(ctx->stream));
    MP_HIP(hipMemcpy(hrec.data(), model->rec, hrec.size() * sizeof(Rec), hipMemcpyDeviceToHost));
    // MP_UCT_COARSE_BITS=n (1..32, default 32): keep only the top n bits in the fused records -- a test knob that
    // makes the exact-row fallback frequent (probability ~|A| * 2^-n per step)
    int shift = 21;
    if (const char *e = getenv("MP_UCT_COARSE_BITS")) {
        const int n = atoi(e);
        if (n >= 1 && n <= 32) shift = 53 - n;
    }
    // record formats: always 32 / 48 bytes with 32 coarse bits per threshold; when |A| <= 5 and S <= 2^20 also 16 bytes
    // (10 coarse bits per threshold in the spare bits of next and flags), which saturated batches use (one gather per
    // rollout step instead of two: +5-9 % there, -5 % on a lone wave).  MP_UCT_POLICY_RECORD=fused / packed forces one.
    const bool can_pack = A <= 5 && S <= (1 << 20) && !getenv("MP_UCT_COARSE_BITS") && !listed && !rollout_slot;
    std::vector<uint32_t> lmask((size_t)S, (1u << A) - 1u); // actions the prior policy lists per state
    if (listed)
        for (int s = 0; s < S; ++s) {
            uint32_t m = 0;
            for (int a = 0; a < A; ++a) m |= (listed[(size_t)s * A + a] ? 1u : 0u) << a;
            if (!m) return fail(MP_ERR_ARG, "mp_policy_load_listed: the prior policy lists no action in state %d", s);
            lmask[(size_t)s] = m;
        }
    std::vector<uint32_t> hf((size_t)S * A * frq * 4, 0xffffffffu), hp16(can_pack ? (size_t)S * A * 4 : 0);
    for (size_t i = 0; i < (size_t)S * A; ++i) {
        uint32_t *f = hf.data() + i * frq * 4;
        memcpy(f, &hrec[i], sizeof(Rec));
        const int nx = hrec[i].next;
        if (nx < 0 || nx >= S) return fail(MP_ERR_ARG, "mp_policy_load: transition out of range");
        f[1] = (hrec[i].flags & 0xffu) | (lmask[(size_t)nx] << 8) | (lmask[i / (size_t)A] << 16);
        uint32_t th[4] = {1023u, 1023u, 1023u, 1023u};
        for (int a = 0; a + 1 < A; ++a) {
            const uint64_t t53 = ht[(size_t)nx * stride + a];
            const uint64_t hi = t53 >> shift;                                      // top bits of the 53, saturated
            f[4 + a] = hi > 0xffffffffULL ? 0xffffffffu : (uint32_t)hi;
            if (can_pack) th[a] = (t53 >> 43) > 1023ULL ? 1023u : (uint32_t)(t53 >> 43);
        }
        if (can_pack) {
            uint32_t *g = hp16.data() + i * 4;
            memcpy(g, &hrec[i], sizeof(Rec));
            g[0] = (uint32_t)nx | (th[0] << 20);
            g[1] = (hrec[i].flags & 3u) | (th[1] << 2) | (th[2] << 12) | (th[3] << 22);
        }
    }
    // records of the rollout, by slot: entry (s, k) = the record of (s, column of slot k) + the next state's thresholds
    std::vector<uint32_t> hfr;
    if (rollout_slot) {
        hfr.resize(hf.size());
        for (int s = 0; s < S; ++s)
            for (int k = 0; k < A; ++k)
                memcpy(hfr.data() + ((size_t)s * A + k) * frq * 4, hf.data() + ((size_t)s * A + col(s, k)) * frq * 4, (size_t)frq * 16);
    }
    mp_policy *pol = new (std::nothrow) mp_policy;
    if (!pol) return fail(MP_ERR_ALLOC, "mp_policy_load: out of memory");
    pol->ctx = ctx; pol->model = model; pol->model_serial = model->serial; pol->S = S; pol->A = A; pol->stride = stride; pol->frq = frq; pol->shift = shift; pol->packed = can_pack ? 1 : 0; pol->listed = listed ? 1 : 0;
    if (hipMalloc(&pol->prior, hp.size() * 8) != hipSuccess || hipMalloc(&pol->thr, ht.size() * 8) != hipSuccess ||
        hipMalloc(&pol->frec, hf.size() * 4) != hipSuccess || hipMalloc(&pol->lmask, lmask.size() * 4) != hipSuccess ||
        (rollout_slot && hipMalloc(&pol->rslot, (size_t)S * A) != hipSuccess) ||
        (rollout_slot && hipMalloc(&pol->frec_roll, hfr.size() * 4) != hipSuccess) ||
        (can_pack && hipMalloc(&pol->frec16, hp16.size() * 4) != hipSuccess)) {
        mp_policy_free(pol);
        return fail(MP_ERR_ALLOC, "mp_policy_load: device allocation failed");
    }
    MP_HIP(hipMemcpy(pol->prior, hp.data(), hp.size() * 8, hipMemcpyHostToDevice));
    MP_HIP(hipMemcpy(pol->thr, ht.data(), ht.size() * 8, hipMemcpyHostToDevice));
    MP_HIP(hipMemcpy(pol->frec, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    // (the same policy also plans through mp_uct_plan_stochastic_policy -- which takes deterministic tables -- e.g. when a
    // binding replays a plan with the visit counter armed: that kernel reads the listed masks and the rollout slots' columns)
    MP_HIP(hipMemcpy(pol->lmask, lmask.data(), lmask.size() * 4, hipMemcpyHostToDevice));
    if (rollout_slot) MP_HIP(hipMemcpy(pol->rslot, rollout_slot, (size_t)S * A, hipMemcpyHostToDevice));
    if (can_pack) MP_HIP(hipMemcpy(pol->frec16, hp16.data(), hp16.size() * 4, hipMemcpyHostToDevice));
    if (rollout_slot) MP_HIP(hipMemcpy(pol->frec_roll, hfr.data(), hfr.size() * 4, hipMemcpyHostToDevice));
    *out = pol;
    return MP_OK;
}

int mp_policy_free(mp_policy *policy)
{
    if (!policy) return MP_OK;
    if (policy->block) { // built on the device: one block of the ctx's cache holds every array
        ctx_block_release(policy->ctx, policy->block, policy->block_bytes);
        delete policy;
        return MP_OK;
    }
    if (policy->prior) (void)hipFree(policy->prior);
    if (policy->thr) (void)hipFree(policy->thr);
    if (policy->frec) (void)hipFree(policy->frec);
    if (policy->frec16) (void)hipFree(policy->frec16);
    if (policy->frec_roll) (void)hipFree(policy->frec_roll);
    if (policy->lmask) (void)hipFree(policy->lmask);
    if (policy->rslot) (void)hipFree(policy->rslot);
    if (policy->listed8) (void)hipFree(policy->listed8);
    delete policy;
    return MP_OK;
}

int mp_uct_step_tree(mp_ctx *ctx, int32_t n_roots, const int32_t *actions, int32_t mem)
{
    if (!ctx || !actions) return fail(MP_ERR_ARG, "mp_uct_step_tree: NULL argument");
    const bool stoch = ctx->tree.kind == 4 && ctx->tree.K == 0; // open-loop trees of mp_uct_plan_stochastic
    if ((ctx->tree.kind != 1 && !stoch) || ctx->tree.n_roots != n_roots)
        return fail(MP_ERR_ARG, "mp_uct_step_tree: no UCT trees of %d roots on this ctx (closed-loop trees cannot be re-used)", n_roots);
    MP_HIP(hipSetDevice(ctx->device));
    // a second step before the next plan (receding_horizon > 1: abstract.py:70-82 steps the tree on every act) descends
    // one more level: apply the pending re-rooting now, then arm the new one
    if (ctx->tree.armed) MP_TRY(stoch ? uct_stoch_reroot_now(ctx, ctx->tree.cap) : uct_reroot_now(ctx, ctx->tree.cap));
    int32_t *d = nullptr;
    MP_TRY(ws_get(ctx, WS_TREE3, (size_t)n_roots, &d));
    MP_HIP(hipMemcpyAsync(d, actions, (size_t)n_roots * sizeof(int32_t),
                          mem == MP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    if (mem == MP_MEM_HOST) MP_HIP(hipStreamSynchronize(ctx->stream));
    ctx->tree.armed = true;
    return MP_OK;
}

int mp_uct_tree_capacity(mp_ctx *ctx, int32_t *cap)
{
    if (!ctx || !cap) return fail(MP_ERR_ARG, "mp_uct_tree_capacity: NULL argument");
    if (ctx->tree.kind != 1) return fail(MP_ERR_ARG, "mp_uct_tree_capacity: no UCT tree on this ctx");
    *cap = ctx->tree.cap;
    return MP_OK;
}

int mp_uct_reset_tree(mp_ctx *ctx)
{
    if (!ctx) return fail(MP_ERR_ARG, "ctx is NULL");
    ctx->tree.armed = false;
    return MP_OK;
}

int mp_uct_path_count(mp_ctx *ctx, int32_t root, const int32_t *actions, int32_t n, int64_t *count)
{
    if (!ctx || !count || (n > 0 && !actions)) return fail(MP_ERR_ARG, "mp_uct_path_count: NULL argument");
    if (ctx->tree.kind != 1) return fail(MP_ERR_ARG, "mp_uct_path_count: no UCT tree on this ctx");
    if (root < 0 || root >= ctx->tree.n_roots || n < 0) return fail(MP_ERR_ARG, "mp_uct_path_count: root %d / length %d out of range", root, n);
    MP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    UctNode *trees = (UctNode *)ctx->ws[ctx->tree.buf ? WS_TREE2 : WS_TREE0].p;
    int32_t *d_act = nullptr;
    int64_t *d_out = nullptr;
    MP_TRY(ws_get(ctx, WS_IO1, (size_t)(n > 0 ? n : 1), &d_act));
    MP_TRY(ws_get(ctx, WS_IO8, 1, &d_out));
    if (n > 0) MP_HIP(hipMemcpyAsync(d_act, actions, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    const int tcap = ctx->tree.cap, A = ctx->tree.A;
    if (ctx->tree.il == 2) hipLaunchKernelGGL(uct_path_count_kernel<2>, dim3(1), dim3(1), 0, st, trees, root, tcap, A, d_act, n, d_out);
    else if (ctx->tree.il == 1) hipLaunchKernelGGL(uct_path_count_kernel<1>, dim3(1), dim3(1), 0, st, trees, root, tcap, A, d_act, n, d_out);
    else hipLaunchKernelGGL(uct_path_count_kernel<0>, dim3(1), dim3(1), 0, st, trees, root, tcap, A, d_act, n, d_out);
    MP_HIP(hipGetLastError());
    MP_HIP(hipMemcpyAsync(count, d_out, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    MP_HIP(hipStreamSynchronize(st));
    return MP_OK;
}

int mp_uct_tree_export(mp_ctx *ctx, int32_t root, int32_t cap, int32_t *n_nodes, int32_t *parent, int32_t *action,
                       int64_t *count, double *value, int32_t *first_child, int32_t *n_children)
{
    if (!ctx) return fail(MP_ERR_ARG, "ctx is NULL");
    if (ctx->tree.kind != 1) return fail(MP_ERR_ARG, "mp_uct_tree_export: no UCT tree on this ctx");
    if (root < 0 || root >= ctx->tree.n_roots) return fail(MP_ERR_ARG, "mp_uct_tree_export: root %d out of range", root);
    const int tcap = ctx->tree.cap, A = ctx->tree.A;
    std::vector<UctNode> h((size_t)tcap);
    MP_HIP(hipSetDevice(ctx->device));
    MP_HIP(hipStreamSynchronize(ctx->stream));
    const UctNode *trees = (const UctNode *)ctx->ws[ctx->tree.buf ? WS_TREE2 : WS_TREE0].p;
    if (ctx->tree.il == 2) {
        // sibling groups of |A| nodes at stride 64 |A|; group 0 holds the root in its last slot, group g >= 1 the ids 1 + (g-1)|A| ..
        const int groups = (tcap - 1) / A + 1;
        std::vector<UctNode> g((size_t)groups * A);
        MP_HIP(hipMemcpy2D(g.data(), (size_t)A * sizeof(UctNode), trees + ((long)(root >> 6) * (tcap + A) * 64 + (long)(root & 63) * A),
                           (size_t)64 * A * sizeof(UctNode), (size_t)A * sizeof(UctNode), (size_t)groups, hipMemcpyDeviceToHost));
        for (int i = 0; i < tcap; ++i) h[i] = g[(size_t)(i + A - 1)];
    } else if (ctx->tree.il == 1) // node i of root r sits at [(r / 64) * cap + i][r % 64]: a strided column
        MP_HIP(hipMemcpy2D(h.data(), sizeof(UctNode), trees + ((long)(root >> 6) * tcap * 64 + (root & 63)), 64 * sizeof(UctNode),
                           sizeof(UctNode), (size_t)tcap, hipMemcpyDeviceToHost));
    else
        MP_HIP(hipMemcpy(h.data(), trees + (long)root * tcap, (size_t)tcap * sizeof(UctNode), hipMemcpyDeviceToHost));
    // node slots are appended A at a time; the tree in use is the closure of first_child links.  Slots with
    // count < 0 are the phantoms of actions a listed policy did not list (uct_kernel<.., MK>): they are not nodes.
    int n = 1;
    for (int i = 0; i < n && i < tcap; ++i)
        if (h[i].first_child >= 0 && h[i].first_child + A > n) n = h[i].first_child + A;
    std::vector<int32_t> id((size_t)n, -1);
    int kept = 0;
    for (int i = 0; i < n; ++i)
        if (h[i].count >= 0) id[i] = kept++;
    if (kept > cap) return fail(MP_ERR_ARG, "mp_uct_tree_export: capacity %d < %d nodes", cap, kept);
    if (parent) parent[0] = -1;
    if (action) action[0] = -1;
    for (int i = 0; i < n; ++i) {
        if (id[i] < 0) continue;
        const int o = id[i];
        if (count) count[o] = h[i].count;
        if (value) value[o] = h[i].value;
        int first = -1, nc = 0;
        if (h[i].first_child >= 0)
            for (int a = 0; a < A; ++a) {
                const int c = id[h[i].first_child + a];
                if (c < 0) continue;
                if (first < 0) first = c;
                ++nc;
                if (parent) parent[c] = o;
                if (action) action[c] = a;
            }
        if (first_child) first_child[o] = first;
        if (n_children) n_children[o] = nc;
    }
    if (n_nodes) *n_nodes = kept;
    return MP_OK;
}

} // extern "C"
