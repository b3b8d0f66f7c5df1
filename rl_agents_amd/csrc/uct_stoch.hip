// uct_stoch.hip -- MCTS / UCT on STOCHASTIC finite MDPs, open and closed loop (reference rl_agents/agents/tree_search/
// mcts.py:132-184, MCTSNode.get_child :267-273; the env side -- next state = rng.choice(n, p=row) with the ENV's own numpy
// generator -- is rl_agents_amd/envs/finite_mdp.py's restatement of the absent `finite_mdp` package).
//
// What is different from uct.hip (deterministic tables):
//   * a transition SAMPLES the next state: one double of the env's PCG64 stream, inverse CDF over the row.  The rows are
//     kept as integer thresholds ceil(cdf * 2^53) (built on the device from the model's probabilities with numpy's
//     arithmetic: sequential cumsum, one IEEE division by the last element): a step is a binary search on uint64 -- or,
//     for rows of at most four successors / non-zero entries, ONE gather of a fused record (below);
//   * every episode steps a deep copy of the env (mcts.py:183), and the copy includes the env's generator: each episode
//     of a plan starts from the SAME env generator record (a second 48-byte record per root, read-only);
//   * closed loop: an action node has one child per DISTINCT next state observed after it, created on first visit, in
//     first-visit order (a linked list: `first` / `next`); the statistics of an observation node are its own.  Nodes are
//     therefore created at data-dependent moments, ids no longer advance in lock-step across the lanes of a wave: the
//     trees are root-major, two 16-byte halves per node with parent links (the export and the oracle walk them literally).
// One root per lane; all randomness of the planner from its numpy-PCG64 record (pcg64.hpp), bit for bit.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "pcg64.hpp"

namespace mp {

// A node is two 16-byte halves in two arrays: what the descent and the backup touch (every episode, for all |A| children of
// every level) and what only the observation lists, the plan and the export read.  The |A| children of a node are 16 |A|
// contiguous bytes of the first array -- one or two cache lines where 32-byte nodes took two or three -- and the halves a
// plan really visits (~200 nodes x 16 B per root) are what has to stay cached.
struct alignas(16) SHot {
    double value;
    int32_t count;
    int32_t first;  // root / observation node (open loop: any node): first of its |A| contiguous action children, -1 = leaf;
                    // action node in closed loop: head of its observation children (linked by `next`)
};
struct alignas(16) SCold {
    int32_t key;    // action id, or the observed next state
    int32_t next;   // next observation sibling (closed loop), -1
    int32_t parent;
    int32_t is_obs;
};
static_assert(sizeof(SHot) == 16 && sizeof(SCold) == 16, "node halves are one dwordx4 each");

struct StochArgs {
    int n_roots, mode, S, A, W, episodes, horizon, cap, closed_loop, done_on_next, max_steps, max_plan_len;
    int table_n;            // counts 1..table_n (+1) have entries in the quotient tables of `tab`
    const int32_t *T;       // deterministic: [S*A]
    const uint64_t *thr;    // dense [S*A][S] / sparse [S*A][B]: ceil(cdf * 2^53)
    const int32_t *nxt;     // sparse: [S*A][B]
    const uint4 *srec;      // sparse, B <= 4: fused records (WB = 2: 2 uint4 per (s, a), WB = 4: 4, WB = 1: 1), else nullptr
    const double *rtab;     // WB = 1: the distinct rewards [256]
    const double *R;        // [S*A]
    const uint8_t *term;    // [S] or nullptr
    const int32_t *root_state, *root_steps;
    const double *tab;      // gpow[H+1] | rollout thresholds [A] (uint64 bits) | tp[A] = (temperature * |A|) * prior[a]
    // per-state policies (mp_policy of a stochastic model): nullptr for the state-independent ones
    const double *pol_prior;   // [S][pol_stride]  prior[s][a], 0 for the actions the prior policy does not list
    const uint64_t *pol_thr;   // [S][pol_stride]  ceil(cdf * 2^53) of the rollout policy, by rollout slot
    const uint32_t *pol_mask;  // [S]  actions the prior policy lists (|A| <= 32)
    const uint8_t *pol_listed; // [S][A] the same as a byte per action (more than 32 actions), else nullptr
    const uint8_t *pol_rslot;  // [S][A] column of rollout slot k, or nullptr (slots are the columns)
    int pol_stride;
    double temperature;
    uint64_t *rng;
    const uint64_t *env_rng;
    SHot *hot;              // [n_roots][cap]
    SCold *cold;            // [n_roots][cap]
    int32_t *visits;           // [n_roots][S] or nullptr: += 1 for the state every env step of the plan lands in (planner.observations)
    const int32_t *n_nodes_in; // kept (re-rooted) tree sizes, nullptr = every root starts fresh (step_strategy "subtree", open loop)
    int32_t *n_nodes_out;
    int32_t *plans, *plan_len;
    double *root_value, *root_child_value;
    int64_t *root_child_count, *env_steps;
};

// numpy: cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(cdf, u, 'right') with u = k * 2^-53:
// #{j : cdf[j] <= u} = #{j : ceil(cdf[j] * 2^53) <= k}.  One thread per (s, a) row.
__global__ void build_thresholds(long rows, int W, const double *__restrict__ P, uint64_t *__restrict__ thr)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const double *p = P + i * W;
    double acc = 0.0;
    for (int j = 0; j < W; ++j) acc += p[j]; // (the running sums below repeat these additions in the same order)
    const double last = acc;
    acc = 0.0;
    for (int j = 0; j < W; ++j) {
        acc += p[j];
        const double scaled = ceil(ldexp(acc / last, 53));
        thr[i * W + j] = scaled >= 18446744073709551615.0 ? ~0ULL : (uint64_t)scaled;
    }
}

// Sparse models with B <= 4 successors: everything an env step reads, in one record per (s, a), so that a step is ONE gather
// instead of the chain threshold(s) -> next state -> terminal flag of the next state (three to four dependent round trips
// of a lone wave).  Only the first B - 1 thresholds can decide a draw (the last one is ceil(1.0 * 2^53) > every k).
//   WB = 2:  q0 = {thr0.lo, thr0.hi, nxt0, nxt1}   q1 = {reward.lo, reward.hi, flags, 0}
//   WB = 4:  q0 = {thr0, thr1}   q1 = {thr2, reward}   q2 = {nxt0..nxt3}   q3 = {flags, 0, 0, 0}
// flags: bit 0 = terminal[s], bit 1 + j = terminal[nxt_j].  Unused slots: threshold 2^64 - 1 (never <= k), last successor.
template <int WB>
__global__ void pack_sparse_records(long rows, int A, int B, const uint64_t *__restrict__ thr, const int32_t *__restrict__ nxt,
                                    const double *__restrict__ R, const uint8_t *__restrict__ term, uint4 *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    uint64_t t[3] = {~0ULL, ~0ULL, ~0ULL};
    int32_t n[4];
    for (int j = 0; j < 4; ++j) n[j] = nxt[i * B + (j < B ? j : B - 1)];
    for (int j = 0; j < B - 1 && j < 3; ++j) t[j] = thr[i * B + j];
    uint32_t flags = term && term[i / A] ? 1u : 0u;
    for (int j = 0; j < 4; ++j) flags |= (term && term[n[j]] ? 1u : 0u) << (1 + j);
    const unsigned long long rb = (unsigned long long)__double_as_longlong(R[i]);
    if (WB == 2) {
        out[i * 2] = make_uint4((uint32_t)t[0], (uint32_t)(t[0] >> 32), (uint32_t)n[0], (uint32_t)n[1]);
        out[i * 2 + 1] = make_uint4((uint32_t)rb, (uint32_t)(rb >> 32), flags, 0u);
    } else {
        out[i * 4] = make_uint4((uint32_t)t[0], (uint32_t)(t[0] >> 32), (uint32_t)t[1], (uint32_t)(t[1] >> 32));
        out[i * 4 + 1] = make_uint4((uint32_t)t[2], (uint32_t)(t[2] >> 32), (uint32_t)rb, (uint32_t)(rb >> 32));
        out[i * 4 + 2] = make_uint4((uint32_t)n[0], (uint32_t)n[1], (uint32_t)n[2], (uint32_t)n[3]);
        out[i * 4 + 3] = make_uint4(flags, 0u, 0u, 0u);
    }
}

// Dense models ([S, A, S] probabilities, the reference's finite-MDP form) whose rows hold at most four non-zero entries take
// the same records: numpy samples a row by cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(cdf, u, 'right'), an entry of
// probability 0 adds exactly 0.0 to the running sum and can never be the first index with cdf > u, so the draw is decided
// by the thresholds of the non-zero entries alone -- the same numbers, computed here in the same order over the full row.
__global__ void dense_row_width(long rows, int W, const double *__restrict__ P, int *__restrict__ max_nnz)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int n = 0;
    if (i < rows)
        for (int j = 0; j < W; ++j) n += P[i * W + j] != 0.0 ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(n, off); n = o > n ? o : n; }
    if ((threadIdx.x & 63) == 0) atomicMax(max_nnz, n);
}

template <int WB>
__global__ void pack_dense_records(long rows, int A, int W, const double *__restrict__ P, const double *__restrict__ R,
                                   const uint8_t *__restrict__ term, uint4 *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const double *p = P + i * W;
    double acc = 0.0;
    for (int j = 0; j < W; ++j) acc += p[j]; // (build_thresholds' additions, in its order)
    const double last = acc;
    uint64_t t[4] = {~0ULL, ~0ULL, ~0ULL, ~0ULL};
    int32_t n[4] = {0, 0, 0, 0};
    int cnt = 0;
    acc = 0.0;
    for (int j = 0; j < W; ++j) {
        acc += p[j];
        if (p[j] != 0.0 && cnt < 4) {
            const double scaled = ceil(ldexp(acc / last, 53));
            t[cnt] = scaled >= 18446744073709551615.0 ? ~0ULL : (uint64_t)scaled;
            n[cnt] = j;
            ++cnt;
        }
    }
    for (int j = cnt; j < 4; ++j) { t[j] = ~0ULL; n[j] = cnt ? n[cnt - 1] : 0; }
    if (cnt) t[cnt - 1] = ~0ULL; // the last non-zero entry's threshold (2^53, or whatever rounding left) is never "<= k": it takes the rest
    uint32_t flags = term && term[i / A] ? 1u : 0u;
    for (int j = 0; j < 4; ++j) flags |= (term && term[n[j]] ? 1u : 0u) << (1 + j);
    const unsigned long long rb = (unsigned long long)__double_as_longlong(R[i]);
    if (WB == 2) {
        out[i * 2] = make_uint4((uint32_t)t[0], (uint32_t)(t[0] >> 32), (uint32_t)n[0], (uint32_t)n[1]);
        out[i * 2 + 1] = make_uint4((uint32_t)rb, (uint32_t)(rb >> 32), flags, 0u);
    } else {
        out[i * 4] = make_uint4((uint32_t)t[0], (uint32_t)(t[0] >> 32), (uint32_t)t[1], (uint32_t)(t[1] >> 32));
        out[i * 4 + 1] = make_uint4((uint32_t)t[2], (uint32_t)(t[2] >> 32), (uint32_t)rb, (uint32_t)(rb >> 32));
        out[i * 4 + 2] = make_uint4((uint32_t)n[0], (uint32_t)n[1], (uint32_t)n[2], (uint32_t)n[3]);
        out[i * 4 + 3] = make_uint4(flags, 0u, 0u, 0u);
    }
}

// Two successors at most AND at most 256 distinct reward values in the model (grid- and highway-like models have a
// handful): the record shrinks to ONE uint4, the reward becomes an 8-bit index into a table staged in LDS -- an env step
// is one 16-byte gather, as in the deterministic kernel, and the count of scattered vector-memory instructions is what
// bounds this kernel at scale (profiles/r03_uct_stoch_units.txt).
//   {thr0.lo, thr0.hi (22 bits) | reward index << 22 | terminal[s] << 30, nxt0 | terminal[nxt0] << 31, nxt1 | terminal[nxt1] << 31}
__global__ void compact_records16(long rows, const uint4 *__restrict__ rec32, const uint8_t *__restrict__ ridx, uint4 *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const uint4 q0 = rec32[i * 2], q1 = rec32[i * 2 + 1];
    const uint32_t flags = q1.z;
    // thr0 <= 2^53 (22 bits of high word) or the "never" mark 2^64 - 1, which becomes 2^54 - 1 (still > every k)
    const uint32_t hi = q0.y > 0x3fffffu ? 0x3fffffu : q0.y;
    out[i] = make_uint4(q0.x, hi | ((uint32_t)ridx[i] << 22) | ((flags & 1u) << 30), q0.z | (((flags >> 1) & 1u) << 31),
                        q0.w | (((flags >> 2) & 1u) << 31));
}

// WB: 0 = any model (deterministic table, dense rows by binary search, sparse rows of any width); 2 / 4 = sparse model through
// the fused records above.  AT: |A| at compile time (2..8: the children of a node in registers -- one batch of loads, scores
// computed once, no loop-carried branches), 0 = any |A|.
// PT: the type of a path-stack entry (uint16_t while every node id fits: the stack is the kernel's LDS footprint, and LDS is
// what limits the waves per SIMD here -- 27 KB per wave held a 262 144-root batch at ONE wave per SIMD).
// SP: per-state prior / rollout policies (restricted action sets, mcts.py:59-97; prior agents, mcts_with_prior.py:47-62).
// A node is expanded with the actions and priors of the state the env is in AT THAT MOMENT (mcts.py:151-154,237-246) and
// keeps them whatever state later episodes reach it in: the prior of a child is STORED (the first 8 bytes of the action
// node's otherwise unused cold half), an unlisted action's slot is a PHANTOM (count = -1: never scored, never visited,
// dropped by the export) and len(children) is the number of real slots of the scored group.
template <int WB, int AT, typename PT, bool SP = false>
__global__ __launch_bounds__(64) void uct_stoch_kernel(StochArgs p)
{
    // (SP with AT == 0: per-state policies over any number of actions -- the loop forms below, round 4)
    extern __shared__ __attribute__((aligned(16))) double lds_s[];
    const int lane = threadIdx.x, A = AT > 0 ? AT : p.A, H = p.horizon, E = p.episodes;
    constexpr int AR = AT > 0 ? AT : 1;
    double *gpow = lds_s;                                   // [H + 1]  gamma ** h
    const uint64_t *rthr = reinterpret_cast<const uint64_t *>(gpow + (H + 1)); // [A] rollout thresholds
    const double *tp = gpow + (H + 1) + A;                  // [A]      temperature * |A| * prior[a]
    const int TE = p.table_n;                               // min(E, 512): counts covered by the quotient tables
    const double *rcp = tp + A;                             // [TE + 1]  1.0 / n
    const double *tpdiv = rcp + (TE + 1);                   // [A][TE+2] temperature * |A| * prior[a] / n
    const int ntab = (H + 1) + 2 * A + (TE + 1) + A * (TE + 2);
    // host tables instead of f64 divisions (the same correctly rounded quotients; a division is ~40 instructions of a lone
    // wave's chain); counts beyond the tables (plans of more than 512 episodes) take the IEEE division itself
    auto explore = [&](int a, int cnt1) { return cnt1 <= TE + 1 ? tpdiv[a * (TE + 2) + cnt1] : tp[a] / (double)cnt1; };
    auto inv = [&](int c) { return c <= TE ? rcp[c] : 1.0 / (double)c; };
    constexpr int NRT = WB == 1 ? 256 : 0;
    double *rtab = lds_s + ntab;                            // [256] WB = 1: the model's distinct rewards
    PT *path = reinterpret_cast<PT *>(lds_s + ntab + NRT) + lane; // entry i of this lane: path[i * 64]
    for (int i = lane; i < ntab; i += 64) lds_s[i] = p.tab[i];
    for (int i = lane; i < NRT; i += 64) rtab[i] = p.rtab[i];
    __syncthreads();
    const int r = blockIdx.x * 64 + lane;
    if (r >= p.n_roots) return;
    SHot *hot = p.hot + (long)r * p.cap;
    SCold *cold = p.cold + (long)r * p.cap;
    Pcg64 g;
    g.load(p.rng + (long)r * 6);
    const int32_t s0 = p.root_state[r], st0 = p.root_steps ? p.root_steps[r] : 0;
    const bool closed = p.closed_loop != 0;
    // (the cold half of an ACTION node is never written: its key is its position in the sibling group, its parent the node
    // whose `first` names the group -- the export derives both; only the root and observation nodes carry one)
    auto make = [&](int id, int parent, int key, int obs) {
        SHot h;
        h.value = 0.0; h.count = 0; h.first = -1;
        hot[id] = h;
        if (obs || parent < 0) {
            SCold c;
            c.key = key; c.next = -1; c.parent = parent; c.is_obs = obs;
            cold[id] = c;
        }
    };
    uint64_t rt[AR]; // the rollout policy's thresholds (wave-uniform)
#pragma unroll
    for (int j = 0; j < AR; ++j) rt[j] = AT > 0 ? rthr[j] : 0ULL;
    int n_nodes = p.n_nodes_in ? p.n_nodes_in[r] : 0; // > 0: a tree kept by step_strategy "subtree" (open loop only)
    if (n_nodes < 1) {
        make(0, -1, -1, 0); // mcts.py:129-130 reset()
        n_nodes = 1;
    }
    long steps_taken = 0;
    // statistics of the first NK path nodes below the root as the descent read them (nobody writes them in between): their
    // backup needs no load (registers, statically indexed -- a load per path node is a scattered vector-memory instruction)
    constexpr int NK = 6;
    double kv[NK];
    int kc[NK];
#pragma unroll
    for (int q = 0; q < NK; ++q) { kv[q] = 0.0; kc[q] = 0; }
    double root_v = 0.0; // the root's statistics live in registers for the plan (every episode reads and updates them)
    int root_c = 0, root_first = -1;
    if (p.n_nodes_in && p.n_nodes_in[r] > 0) { // (a kept root continues from its statistics)
        const SHot h0 = hot[0];
        root_v = h0.value; root_c = h0.count; root_first = h0.first;
    }
#ifdef MP_PROFILE
    long long t_sel = 0, t_exp = 0, t_roll = 0, t_bak = 0, n_sel = 0, n_roll = 0;
    const long long t_all0 = clock64();
#define SPROF(x) const long long x = clock64()
#else
#define SPROF(x)
#endif
    for (int ep = 0; ep < E; ++ep) { // mcts.py:179-184
        SPROF(c0);
        int32_t s = s0, st = st0;    // safe_deepcopy_env(state): the clone's state, step counter ...
        Pcg64 eg;                    // ... and a COPY of the env's generator: every episode replays the same noise
        eg.s_hi = eg.s_lo = eg.inc_hi = 0; eg.inc_lo = 1; eg.has_uint32 = eg.uinteger = 0;
        if (p.env_rng) eg.load(p.env_rng + (long)r * 6); // (a deterministic model draws nothing from it)
        // one env.step(a): -> reward, terminated, truncated; advances (s, st) and, for a stochastic model, eg
        auto env_step = [&](int a, double &reward, bool &terminated, bool &truncated) {
            const long sa = (long)s * A + a;
            int32_t sn;
            if (WB == 1) {
                const uint4 q = p.srec[sa];
                const uint64_t k = eg.next64() >> 11; // Generator.random(): runs while the gather is in flight
                const bool up = ((((uint64_t)(q.y & 0x3fffffu)) << 32) | q.x) <= k; // searchsorted(cdf, u, 'right')
                const uint32_t nw = up ? q.w : q.z;
                reward = rtab[(q.y >> 22) & 0xffu];
                terminated = p.done_on_next ? (nw >> 31) != 0 : ((q.y >> 30) & 1u) != 0;
                s = (int32_t)(nw & 0x7fffffffu);
                st += 1;
                truncated = p.max_steps > 0 && st >= p.max_steps;
                ++steps_taken;
                return;
            }
            if (WB > 0) {
                // the draw does not depend on the record: its 128-bit multiply runs while the gather is in flight
                const uint4 *rp = p.srec + sa * WB;
                const uint4 q0 = rp[0], q1 = rp[1];
                uint4 q2 = make_uint4(0u, 0u, 0u, 0u), q3 = q2;
                if (WB == 4) { q2 = rp[2]; q3 = rp[3]; }
                const uint64_t k = eg.next64() >> 11; // Generator.random()
                int lo;                               // searchsorted(cdf, u, 'right') = #{j : thr_j <= k}
                uint32_t flags;
                if (WB == 2) {
                    lo = (((uint64_t)q0.y << 32) | q0.x) <= k ? 1 : 0;
                    sn = (int32_t)(lo ? q0.w : q0.z);
                    reward = __hiloint2double((int)q1.y, (int)q1.x);
                    flags = q1.z;
                } else {
                    lo = ((((uint64_t)q0.y << 32) | q0.x) <= k ? 1 : 0) + ((((uint64_t)q0.w << 32) | q0.z) <= k ? 1 : 0) +
                         ((((uint64_t)q1.y << 32) | q1.x) <= k ? 1 : 0);
                    sn = (int32_t)(lo == 0 ? q2.x : lo == 1 ? q2.y : lo == 2 ? q2.z : q2.w);
                    reward = __hiloint2double((int)q1.w, (int)q1.z);
                    flags = q3.x;
                }
                terminated = p.done_on_next ? ((flags >> (1 + lo)) & 1u) != 0 : (flags & 1u) != 0;
                s = sn;
                st += 1;
                truncated = p.max_steps > 0 && st >= p.max_steps;
                ++steps_taken;
                return;
            }
            if (p.mode == MP_MODE_DETERMINISTIC) {
                sn = p.T[sa];
            } else {
                const uint64_t k = eg.next64() >> 11; // Generator.random()
                const uint64_t *row = p.thr + sa * p.W;
                int lo = 0, hi = p.W;                 // searchsorted(cdf, u, 'right')
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (row[mid] <= k) lo = mid + 1; else hi = mid;
                }
                if (lo >= p.W) lo = p.W - 1;          // (u < 1 = cdf[-1]: not reached)
                sn = p.mode == MP_MODE_SPARSE ? p.nxt[sa * p.W + lo] : lo;
            }
            reward = p.R[sa];
            terminated = p.term ? (p.done_on_next ? p.term[sn] != 0 : p.term[s] != 0) : false;
            s = sn;
            st += 1;
            truncated = p.max_steps > 0 && st >= p.max_steps;
            ++steps_taken;
        };
        int node = 0, depth = 0, plen = 0;
        bool terminal = false;
        double total = 0.0;
        path[(plen++) * 64] = (PT)0;
        int fc = root_first;
        // ---- selection, mcts.py:143-149
        while (depth < H && fc >= 0 && !terminal) {
            // MCTSNode.selection_strategy (:275-286): value + temperature * len(children) * prior / (count + 1);
            // Node.random_argmax (abstract.py:296-311): exact-equality argmax set, one bounded draw among >= 2 ties
            int act = 0, act_first = -1, act_c = 0;
            double act_v = 0.0;
            if (AT > 0) {
                SHot c[AR];
#pragma unroll
                for (int a = 0; a < AR; ++a) c[a] = hot[fc + a];
                double sc[AR];
                if (SP) {
                    double pr[AR];
                    int nl = 0;
#pragma unroll
                    for (int a = 0; a < AR; ++a) {
                        pr[a] = *reinterpret_cast<const double *>(&cold[fc + a]);
                        nl += c[a].count >= 0 ? 1 : 0;
                    }
                    const double TAk = p.temperature * (double)nl; // mcts.py:286, left to right
#pragma unroll
                    for (int a = 0; a < AR; ++a)
                        sc[a] = c[a].count < 0 ? -INFINITY : c[a].value + (TAk * pr[a]) / (double)(c[a].count + 1);
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
#pragma unroll
                for (int a = 0; a < AR; ++a) {
                    const bool eq = sc[a] == m;
                    if (eq && !found && pick == 0) {
                        act = a; found = true;
                        act_first = c[a].first; act_c = c[a].count; act_v = c[a].value;
                    }
                    if (eq && !found) --pick;
                }
            } else {
                double TAk = 0.0;
                if (SP) { // len(children) = the listed slots of this group (a phantom slot has count < 0)
                    int nl = 0;
                    for (int a = 0; a < A; ++a) nl += hot[fc + a].count >= 0 ? 1 : 0;
                    TAk = p.temperature * (double)nl; // mcts.py:286, left to right
                }
                auto score = [&](const SHot &c, int a) -> double {
                    if (SP) {
                        if (c.count < 0) return -INFINITY;
                        const double pr = *reinterpret_cast<const double *>(&cold[fc + a]); // the prior stored at expansion
                        return c.value + (TAk * pr) / (double)(c.count + 1);
                    }
                    return c.value + explore(a, c.count + 1);
                };
                double m = 0.0;
                int nt = 0;
                for (int a = 0; a < A; ++a) {
                    const SHot c = hot[fc + a];
                    const double sc = score(c, a);
                    if (a == 0 || sc > m) { m = sc; nt = 1; } else if (sc == m) ++nt;
                }
                int pick = nt > 1 ? (int)g.below((uint32_t)nt) : 0;
                for (int a = 0; a < A; ++a) {
                    const SHot c = hot[fc + a];
                    const double sc = score(c, a);
                    if (sc == m) {
                        if (pick == 0) { act = a; act_first = c.first; act_c = c.count; act_v = c.value; break; }
                        --pick;
                    }
                }
            }
            // closed loop: both halves of the head of the action node's observation list are requested together with the env
            // step's record (all three only need the action)
            SHot oh;
            oh.value = 0.0; oh.count = 0; oh.first = -1;
            SCold oc;
            oc.key = -1; oc.next = -1; oc.parent = -1; oc.is_obs = 1;
            if (closed && act_first >= 0) { oc = cold[act_first]; oh = hot[act_first]; }
            double reward;
            bool trunc;
            env_step(act, reward, terminal, trunc);
            if (p.visits) atomicAdd(p.visits + (long)r * p.S + s, 1); // AbstractPlanner.step: observations.append (abstract.py:158-161)
            total += gpow[depth] * reward;
            node = fc + act;
#pragma unroll
            for (int q = 0; q < NK; ++q)
                if (plen == q + 1) { kv[q] = act_v; kc[q] = act_c; }
            path[(plen++) * 64] = (PT)node;
            fc = act_first;
            if (closed) { // get_child(action, observation): the child keyed by str(observation), made on first visit
                int o = act_first, prev = -1;
                while (o >= 0) {
                    if (oc.key == s) break;
                    prev = o; o = oc.next;
                    if (o >= 0) { oc = cold[o]; oh = hot[o]; }
                }
                if (o < 0) {
                    o = n_nodes++;
                    make(o, node, s, 1);
                    if (prev < 0) hot[node].first = o; else cold[prev].next = o;
                    oh.value = 0.0; oh.count = 0; oh.first = -1;
                }
                node = o;
#pragma unroll
                for (int q = 0; q < NK; ++q)
                    if (plen == q + 1) { kv[q] = oh.value; kc[q] = oh.count; }
                path[(plen++) * 64] = (PT)node;
                fc = oh.first;
            }
            ++depth;
        }
        SPROF(c1);
#ifdef MP_PROFILE
        n_sel += depth;
#endif
        // ---- expansion, mcts.py:151-154 (a child per action, prior = the state-independent prior policy's)
        if (fc < 0 && depth < H && (!terminal || node == 0)) {
            const int c0 = n_nodes;
            if (SP) { // prior_policy(state, observation) of the state the clone is in now
                const uint32_t mask = p.pol_mask[s];
                const uint8_t *lrow = p.pol_listed ? p.pol_listed + (long)s * A : nullptr; // more than 32 actions: a byte per action
                const double *row = p.pol_prior + (long)s * p.pol_stride;
                for (int a = 0; a < A; ++a) {
                    SHot h;
                    const bool is_listed = lrow ? lrow[a] != 0 : ((mask >> (a & 31)) & 1u) != 0;
                    h.value = 0.0; h.count = is_listed ? 0 : -1; h.first = -1;
                    hot[c0 + a] = h;
                    *reinterpret_cast<double *>(&cold[c0 + a]) = row[a];
                }
            } else {
                for (int a = 0; a < A; ++a) make(c0 + a, node, a, 0);
            }
            if (node == 0) root_first = c0;
            hot[node].first = c0;
            n_nodes += A;
        }
        // ---- rollout, mcts.py:160-177
        SPROF(c2);
        if (!terminal && depth < H) {
            // The draw of step h + 1 is computed while step h's lookup is in flight -- on a copy of the generator that is
            // committed only if the rollout goes on, so a rollout that stops leaves the stream where the reference's is.
            Pcg64 gn = g;
            uint64_t k = gn.next64() >> 11;          // np_random.choice(actions, 1, p=p): one double, inverse cdf
            for (int h = depth; h < H; ++h) {
                int a = 0;
                if (SP) { // rollout_policy(state, observation): the thresholds of the state the clone is in
                    const uint64_t *tr = p.pol_thr + (long)s * p.pol_stride;
                    if (AT > 0) {
#pragma unroll
                        for (int j = 0; j < AR - 1; ++j) a += tr[j] <= k ? 1 : 0;
                    } else {
                        for (int j = 0; j < A - 1; ++j) a += tr[j] <= k ? 1 : 0;
                    }
                    if (p.pol_rslot) a = p.pol_rslot[(long)s * A + a];
                } else if (AT > 0) {
#pragma unroll
                    for (int j = 0; j < AR - 1; ++j) a += rt[j] <= k ? 1 : 0;
                } else {
                    for (int j = 0; j < A - 1; ++j) a += rthr[j] <= k ? 1 : 0;
                }
                g = gn;                              // this step's draw is consumed
                Pcg64 gs = gn;
                const uint64_t k_next = gs.next64() >> 11;
                double reward;
                bool term_h, trunc_h;
                env_step(a, reward, term_h, trunc_h);
                if (p.visits) atomicAdd(p.visits + (long)r * p.S + s, 1);
                total += gpow[h] * reward;
                if (term_h || trunc_h) break;
                gn = gs;
                k = k_next;
            }
        }
        SPROF(c3);
        // ---- update_branch, mcts.py:248-265: the same total on every node of the path
        {
            const int c = root_c + 1;
            root_c = c;
            root_v = root_v + inv(c) * (total - root_v);
        }
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
#pragma unroll
        for (int q = 0; q < NK; ++q)
            if (q + 1 < plen) {
                SHot *nd = hot + (int)path[(q + 1) * 64];
                const int c = kc[q] + 1;
                const double nv = kv[q] + inv(c) * (total - kv[q]);
                u32x3 w = {(uint32_t)__double2loint(nv), (uint32_t)__double2hiint(nv), (uint32_t)c};
                *reinterpret_cast<u32x3 *>(nd) = w;
            }
        for (int i = NK + 1; i < plen; ++i) {
            SHot *nd = hot + (int)path[i * 64];
            const int c = nd->count + 1;
            const double v = nd->value;
            const double nv = v + inv(c) * (total - v);
            // {value, count} as ONE 12-byte store (two stores were two scattered vector-memory instructions)
            u32x3 w = {(uint32_t)__double2loint(nv), (uint32_t)__double2hiint(nv), (uint32_t)c};
            *reinterpret_cast<u32x3 *>(nd) = w;
        }
#ifdef MP_PROFILE
        { const long long c4 = clock64(); t_sel += c1 - c0; t_exp += c2 - c1; t_roll += c3 - c2; t_bak += c4 - c3; }
#endif
    }
#ifdef MP_PROFILE
    if (r == 0)
        printf("uct_stoch prof root0: total=%lld selection=%lld (%lld levels) expansion=%lld rollout=%lld (%lld env steps in all) backup=%lld (clock64 ticks)\n",
               (long long)(clock64() - t_all0), t_sel, n_sel, t_exp, t_roll, (long long)steps_taken, t_bak);
#endif
    hot[0].value = root_v;
    hot[0].count = root_c;
    g.store(p.rng + (long)r * 6);
    if (p.n_nodes_out) p.n_nodes_out[r] = n_nodes;
    if (p.env_steps) p.env_steps[r] = steps_taken;
    if (p.root_value) p.root_value[r] = root_v;
    {
        const int fc = root_first;
        for (int a = 0; a < A; ++a) {
            if (p.root_child_count) p.root_child_count[(long)r * A + a] = fc >= 0 ? max(hot[fc + a].count, 0) : 0;
            if (p.root_child_value) p.root_child_value[(long)r * A + a] = fc >= 0 ? hot[fc + a].value : 0.0;
        }
    }
    // ---- get_plan (abstract.py:143-156) with MCTSNode.selection_rule (mcts.py:212-218) at every level: most visited
    // child, ties to the first largest value; under an action node of a closed-loop tree the "action" is the observation key
    int len = 0, node = 0;
    bool at_action_node = false;
    for (;;) {
        int best = -1, bc = 0, bkey = -1;
        double bv = 0.0;
        if (closed && at_action_node) {
            for (int o = hot[node].first; o >= 0;) {
                const SHot on = hot[o];
                const SCold oc = cold[o];
                if (best < 0 || on.count > bc || (on.count == bc && on.value > bv)) { best = o; bc = on.count; bv = on.value; bkey = oc.key; }
                o = oc.next;
            }
        } else {
            const int fc = hot[node].first;
            if (fc >= 0)
                for (int a = 0; a < A; ++a) {
                    const SHot cn = hot[fc + a];
                    if (cn.count < 0) continue; // (the phantom slot of an unlisted action)
                    if (best < 0 || cn.count > bc || (cn.count == bc && cn.value > bv)) { best = fc + a; bc = cn.count; bv = cn.value; bkey = a; }
                }
        }
        if (best < 0) break;
        if (p.plans && len < p.max_plan_len) p.plans[(long)r * p.max_plan_len + len] = bkey;
        ++len;
        node = best;
        at_action_node = !at_action_node;
    }
    if (p.plans)
        for (int i = len; i < p.max_plan_len; ++i) p.plans[(long)r * p.max_plan_len + i] = -1;
    if (p.plan_len) p.plan_len[r] = len;
}

// AbstractPlanner.step_by_subtree (abstract.py:195-206) on OPEN-LOOP trees of this kernel, one root per lane: the subtree of
// the root's child `action` is re-numbered breadth-first into the other buffer (every expanded node's |A| children stay
// contiguous).  While a node waits in the queue its `first` holds its OLD id.  A never-expanded root gives size 0.
// old_cold != nullptr (per-state policies): the action nodes' cold halves hold their stored priors and travel with them.
__global__ __launch_bounds__(64) void uct_stoch_reroot_kernel(int n_roots, int A, int cap_old, int cap_new,
                                                              const SHot *__restrict__ old_hot, SHot *__restrict__ new_hot,
                                                              const SCold *__restrict__ old_cold, SCold *__restrict__ cold,
                                                              const int32_t *n_old, const int32_t *__restrict__ actions, int32_t *n_new)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_roots) return;
    const SHot *o = old_hot + (long)r * cap_old;
    SHot *n = new_hot + (long)r * cap_new;
    const SCold *oc = old_cold ? old_cold + (long)r * cap_old : nullptr;
    SCold *nc = cold + (long)r * cap_new;
    const int a = actions[r];
    // `if action in self.root.children` (abstract.py:201): an unlisted action has a phantom slot, not a child
    if (n_old[r] < 1 || o[0].first < 0 || a < 0 || a >= A || o[o[0].first + a].count < 0) {
        n_new[r] = 0;
        return;
    }
    int head = 0, tail = 1;
    SHot first;
    first.value = 0.0; first.count = 0; first.first = o[0].first + a;
    n[0] = first;
    while (head < tail) {
        const SHot src = o[n[head].first];
        SHot out;
        out.value = src.value; out.count = src.count; out.first = -1;
        if (src.first >= 0) {
            out.first = tail;
            for (int c = 0; c < A; ++c) {
                SHot q;
                q.value = 0.0; q.count = 0; q.first = src.first + c;
                n[tail + c] = q;
                if (oc) nc[tail + c] = oc[src.first + c];
            }
            tail += A;
        }
        n[head] = out;
        ++head;
    }
    SCold c0;
    c0.key = -1; c0.next = -1; c0.parent = -1; c0.is_obs = 0;
    nc[0] = c0; // (the root is the only node of an open-loop tree whose cold half is a record)
    n_new[r] = tail;
}

// The env side of Evaluation.step (trainer/evaluation.py:164-190) for n lock-step episodes of a STOCHASTIC model: as
// env_step_kernel (api.hip), with the next state sampled from the episode's OWN env generator (advanced in place), exactly
// as FiniteMDPEnv.step does: one Generator.random() double, inverse CDF over the row's integer thresholds.
__global__ void env_step_stoch_kernel(int n, int A, int sparse, int W, const uint64_t *__restrict__ thr, const int32_t *__restrict__ nxt,
                                      const double *__restrict__ R, const uint8_t *__restrict__ term, int done_on_next,
                                      int32_t *__restrict__ state, int32_t *__restrict__ steps, uint8_t *__restrict__ alive,
                                      const int32_t *__restrict__ plans, int plan_stride, int max_steps, const double *__restrict__ gpow,
                                      double *__restrict__ returns, double *__restrict__ discounted, int32_t *__restrict__ actions_log,
                                      int log_stride, int32_t *__restrict__ n_alive, uint64_t *__restrict__ env_rng)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = false;
    if (i < n && alive[i]) {
        const int planned = plans[(long)i * plan_stride];
        const int act = planned < 0 ? 0 : planned;
        const int s = state[i], t = steps[i];
        const long sa = (long)s * A + act;
        Pcg64 eg;
        eg.load(env_rng + (long)i * 6);
        const uint64_t k = eg.next64() >> 11; // Generator.random()
        eg.store(env_rng + (long)i * 6);
        const uint64_t *row = thr + sa * W;
        int lo = 0, hi = W;                   // searchsorted(cdf, u, 'right')
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (row[mid] <= k) lo = mid + 1; else hi = mid;
        }
        if (lo >= W) lo = W - 1;
        const int sn = sparse ? nxt[sa * W + lo] : lo;
        const double reward = R[sa];
        const bool done = term ? (done_on_next ? term[sn] != 0 : term[s] != 0) : false;
        returns[i] += reward;
        discounted[i] += reward * gpow[t];
        if (actions_log && t < log_stride) actions_log[(long)i * log_stride + t] = planned < 0 ? -1 : act;
        state[i] = sn;
        steps[i] = t + 1;
        live = !(done || t + 1 >= max_steps);
        alive[i] = live ? 1 : 0;
    }
    const unsigned long long b = __ballot(live);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_alive, (int)__popcll(b));
}

} // namespace mp

using namespace mp;

// sampling thresholds of a stochastic / sparse model's rows, built once per model on the device (numpy's cumsum / division order)
static int ensure_thresholds(mp_ctx *ctx, mp_model *model)
{
    if (model->thr) return MP_OK;
    const long rows = (long)model->S * model->A;
    const int W = model->mode == MP_MODE_STOCHASTIC ? model->S : model->B;
    if (hipMalloc(&model->thr, (size_t)rows * W * sizeof(uint64_t)) != hipSuccess)
        return fail(MP_ERR_ALLOC, "%zu B for the sampling thresholds", (size_t)rows * W * 8);
    hipLaunchKernelGGL(build_thresholds, dim3((unsigned)((rows + 127) / 128)), dim3(128), 0, ctx->stream, rows, W, model->P, model->thr);
    MP_HIP(hipGetLastError());
    return MP_OK;
}

// Apply the re-rooting armed by mp_uct_step_tree to the open-loop trees of the last mp_uct_plan_stochastic: every kept tree
// -> the subtree under its root's child actions[i], into the other hot buffer with node stride cap_new.
int uct_stoch_reroot_now(mp_ctx *ctx, long cap_new)
{
    const int n_roots = ctx->tree.n_roots, A = ctx->tree.A;
    const int old_slot = ctx->tree.buf ? WS_TREE5 : WS_TREE0, new_slot = ctx->tree.buf ? WS_TREE0 : WS_TREE5;
    // (per-state policies keep stored priors in the cold halves: those alternate between two buffers like the hot ones)
    const bool sp = ctx->tree.sp;
    const int old_cold = sp && ctx->tree.buf ? WS_TREE6 : WS_TREE2, new_cold = sp && !ctx->tree.buf ? WS_TREE6 : WS_TREE2;
    SHot *nw = nullptr;
    SCold *cold = nullptr;
    MP_TRY(ws_get(ctx, new_slot, (size_t)n_roots * cap_new, &nw));
    MP_TRY(ws_get(ctx, new_cold, (size_t)n_roots * cap_new, &cold));
    int32_t *sizes = (int32_t *)ctx->ws[WS_TREE1].p;
    const int32_t *acts = (const int32_t *)ctx->ws[WS_TREE3].p;
    hipLaunchKernelGGL(uct_stoch_reroot_kernel, dim3((unsigned)((n_roots + 63) / 64)), dim3(64), 0, ctx->stream, n_roots, A,
                       ctx->tree.cap, (int)cap_new, (const SHot *)ctx->ws[old_slot].p, nw,
                       sp ? (const SCold *)ctx->ws[old_cold].p : (const SCold *)nullptr, cold, sizes, acts, sizes);
    MP_HIP(hipGetLastError());
    ctx->tree.buf ^= 1;
    ctx->tree.cap = (int)cap_new;
    ctx->tree.armed = false;
    return MP_OK;
}

extern "C" {

int mp_model_set_episode_rules(mp_model *model, int32_t done_on_next, int32_t max_steps)
{
    if (!model) return fail(MP_ERR_ARG, "mp_model_set_episode_rules: model is NULL");
    model->done_on_next = done_on_next ? 1 : 0;
    model->max_steps = max_steps > 0 ? max_steps : 0;
    return MP_OK;
}

} // extern "C"

static int uct_stoch_plan_impl(mp_ctx *ctx, mp_model *model, const mp_policy *pol, int32_t n_roots, const int32_t *root_state,
                               const int32_t *root_steps, int32_t episodes, int32_t horizon, double gamma, double temperature,
                               const double *prior_p, const double *rollout_p, int32_t closed_loop, uint64_t *rng_state,
                               const uint64_t *env_rng_state, int32_t max_plan_len, int32_t *plans, int32_t *plan_len,
                               double *root_value, int64_t *root_child_count, double *root_child_value, int64_t *env_steps,
                               int32_t mem)
{
    if (!ctx || !model || !root_state || !rng_state || (!pol && (!prior_p || !rollout_p)))
        return fail(MP_ERR_ARG, "mp_uct_plan_stochastic: NULL argument");
    if (pol && (pol->model != model || pol->model_serial != model->serial || pol->ctx != ctx || !pol->lmask))
        return fail(MP_ERR_ARG, "mp_uct_plan_stochastic_policy: the policy was not loaded for this (stochastic) model");
    if (!mem_valid(mem)) return fail(MP_ERR_ARG, "mp_uct_plan_stochastic: unknown mem flags %d", mem);
    const int mode = model->mode;
    if (mode != MP_MODE_DETERMINISTIC && mode != MP_MODE_STOCHASTIC && mode != MP_MODE_SPARSE)
        return fail(MP_ERR_MODE, "mp_uct_plan_stochastic: model mode %d is not a finite MDP", mode);
    if (mode != MP_MODE_DETERMINISTIC && !env_rng_state)
        return fail(MP_ERR_ARG, "mp_uct_plan_stochastic: a stochastic model needs the env's generator records");
    if (mode == MP_MODE_STOCHASTIC && (model->M != 1 || model->Sc != model->S))
        return fail(MP_ERR_MODE, "mp_uct_plan_stochastic: one full dense model [S,A,S] expected");
    if (model->masked && !pol)
        return fail(MP_ERR_ARG, "mp_uct_plan_stochastic: the model restricts its action sets: plan with a policy (mp_uct_plan_stochastic_policy)");
    if (n_roots < 1 || episodes < 0 || horizon < 0 || max_plan_len < 0)
        return fail(MP_ERR_ARG, "mp_uct_plan_stochastic: bad sizes (n_roots=%d episodes=%d horizon=%d)", n_roots, episodes, horizon);
    const int A = model->A, S = model->S, H = horizon, E = episodes;
    const int W = mode == MP_MODE_STOCHASTIC ? S : (mode == MP_MODE_SPARSE ? model->B : 0);
    MP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int amem = mem_arrays(mem), rmem = mem_rng(mem);

    // Fused records (one gather per env step) for sparse rows of at most four successors and for dense rows with at most
    // four non-zero entries; MP_UCT_STOCH_FUSED=0: never -- test hook.
    const char *fz = getenv("MP_UCT_STOCH_FUSED");
    const bool fuse_ok = !(fz && fz[0] == '0');
    const long rows = (long)S * A;
    if (model->srec_wb == 0 && mode == MP_MODE_SPARSE) model->srec_wb = W <= 2 ? 2 : (W <= 4 ? 4 : -1);
    if (model->srec_wb == 0 && mode == MP_MODE_STOCHASTIC && fuse_ok) {
        int *d_max = nullptr, h_max = 0;
        MP_TRY(ws_get(ctx, WS_IO9, 1, &d_max));
        MP_HIP(hipMemsetAsync(d_max, 0, sizeof(int), st));
        hipLaunchKernelGGL(dense_row_width, dim3((unsigned)((rows + 127) / 128)), dim3(128), 0, st, rows, W, model->P, d_max);
        MP_HIP(hipMemcpyAsync(&h_max, d_max, sizeof(int), hipMemcpyDeviceToHost, st));
        MP_HIP(hipStreamSynchronize(st));
        model->srec_wb = h_max <= 2 ? 2 : (h_max <= 4 ? 4 : -1);
    }
    const int wb = fuse_ok && model->srec_wb > 0 ? model->srec_wb : 0;
    // sampling thresholds of the model's rows (rows that are not fused), built once per model on the device (numpy's cumsum /
    // division order)
    if (mode != MP_MODE_DETERMINISTIC && !model->thr && (!wb || mode == MP_MODE_SPARSE)) {
        if (hipMalloc(&model->thr, (size_t)rows * W * sizeof(uint64_t)) != hipSuccess)
            return fail(MP_ERR_ALLOC, "mp_uct_plan_stochastic: %zu B for the sampling thresholds", (size_t)rows * W * 8);
        hipLaunchKernelGGL(build_thresholds, dim3((unsigned)((rows + 127) / 128)), dim3(128), 0, st, rows, W, model->P, model->thr);
        MP_HIP(hipGetLastError());
    }
    if (wb && !model->srec) {
        if (hipMalloc(&model->srec, (size_t)rows * wb * sizeof(uint4)) != hipSuccess)
            return fail(MP_ERR_ALLOC, "mp_uct_plan_stochastic: %zu B for the fused records", (size_t)rows * wb * sizeof(uint4));
        const dim3 grid((unsigned)((rows + 127) / 128)), block(128);
        if (mode == MP_MODE_SPARSE && wb == 2)
            hipLaunchKernelGGL(pack_sparse_records<2>, grid, block, 0, st, rows, A, W, model->thr, model->NXT, model->R, model->term, model->srec);
        else if (mode == MP_MODE_SPARSE)
            hipLaunchKernelGGL(pack_sparse_records<4>, grid, block, 0, st, rows, A, W, model->thr, model->NXT, model->R, model->term, model->srec);
        else if (wb == 2)
            hipLaunchKernelGGL(pack_dense_records<2>, grid, block, 0, st, rows, A, W, model->P, model->R, model->term, model->srec);
        else
            hipLaunchKernelGGL(pack_dense_records<4>, grid, block, 0, st, rows, A, W, model->P, model->R, model->term, model->srec);
        MP_HIP(hipGetLastError());
        // two successors at most: one uint4 per record when the model has at most 256 distinct rewards (MP_UCT_STOCH_FUSED=2:
        // keep the 32-byte records -- test hook)
        if (wb == 2 && !(fz && fz[0] == '2')) {
            std::vector<double> hr((size_t)rows);
            MP_HIP(hipMemcpyAsync(hr.data(), model->R, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, st));
            MP_HIP(hipStreamSynchronize(st));
            std::vector<uint64_t> vals;  // distinct bit patterns, in order of first appearance
            std::unordered_map<uint64_t, int> index_of;
            std::vector<uint8_t> ridx((size_t)rows);
            bool few = true;
            for (long i = 0; i < rows && few; ++i) {
                uint64_t b;
                memcpy(&b, &hr[(size_t)i], sizeof(b));
                auto it = index_of.find(b);
                if (it == index_of.end()) {
                    if (vals.size() == 256) { few = false; break; }
                    it = index_of.emplace(b, (int)vals.size()).first;
                    vals.push_back(b);
                }
                ridx[(size_t)i] = (uint8_t)it->second;
            }
            if (few) {
                std::vector<double> tabv(256, 0.0);
                for (size_t j = 0; j < vals.size(); ++j) memcpy(&tabv[j], &vals[j], sizeof(double));
                // (the model is touched only after every step has succeeded: a failure leaves the 32-byte records in use)
                uint4 *rec16 = nullptr;
                uint8_t *d_ridx = nullptr;
                double *d_rtab = nullptr;
                auto drop = [&]() { if (rec16) hipFree(rec16); if (d_ridx) hipFree(d_ridx); if (d_rtab) hipFree(d_rtab); };
                if (hipMalloc(&rec16, (size_t)rows * sizeof(uint4)) != hipSuccess || hipMalloc(&d_ridx, (size_t)rows) != hipSuccess ||
                    hipMalloc(&d_rtab, 256 * sizeof(double)) != hipSuccess) {
                    drop();
                    return fail(MP_ERR_ALLOC, "mp_uct_plan_stochastic: the compact records");
                }
                if (hipMemcpyAsync(d_ridx, ridx.data(), (size_t)rows, hipMemcpyHostToDevice, st) != hipSuccess ||
                    hipMemcpyAsync(d_rtab, tabv.data(), 256 * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) {
                    drop();
                    return fail(MP_ERR_HIP, "mp_uct_plan_stochastic: upload of the compact records failed");
                }
                hipLaunchKernelGGL(compact_records16, grid, block, 0, st, rows, model->srec, d_ridx, rec16);
                if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { // (the host vectors go away)
                    drop();
                    return fail(MP_ERR_HIP, "mp_uct_plan_stochastic: compact_records16 failed");
                }
                hipFree(d_ridx);
                hipFree(model->srec);
                model->srec = rec16;
                model->srec_rtab = d_rtab;
                model->srec_wb = 1;
            }
        }
    }
    const int wbk = wb ? model->srec_wb : 0; // the form the records really have (1: compact)

    // per-call tables, computed on the host exactly as Python computes them (see uct_plan_impl)
    int TE = E < 512 ? E : 512; // the quotient tables live in LDS: longer plans divide beyond them
    // (any number of actions: the [A][TE + 2] table of the state-independent prior must leave room for the path stack; counts
    // beyond the tables take the IEEE division itself -- the same quotients)
    while (TE > 1 && (size_t)A * (TE + 2) * sizeof(double) > 24 * 1024) TE >>= 1;
    const size_t ntab = (size_t)(H + 1) + 2 * (size_t)A + (size_t)(TE + 1) + (size_t)A * (TE + 2);
    std::vector<double> tab(ntab);
    double *gpow = tab.data(), *cdf = gpow + (H + 1), *tpv = cdf + A, *rcp = tpv + A, *tpdiv = rcp + (TE + 1);
    for (int h = 0; h <= H; ++h) gpow[h] = pow(gamma, (double)h);
    double acc = 0.0;
    for (int a = 0; a < A; ++a) { acc += pol ? 1.0 : rollout_p[a]; cdf[a] = acc; }
    for (int a = 0; a < A; ++a) {
        const double scaled = ceil(ldexp(cdf[a] / acc, 53));
        const uint64_t t = scaled >= 18446744073709551615.0 ? ~0ULL : (uint64_t)scaled;
        memcpy(&cdf[a], &t, sizeof(t));
    }
    for (int a = 0; a < A; ++a) tpv[a] = temperature * (double)A * (pol ? 0.0 : prior_p[a]); // mcts.py:286, left to right
    rcp[0] = 0.0;
    for (int n = 1; n <= TE; ++n) rcp[n] = 1.0 / (double)n;                    // mcts.py:257: 1 / count
    for (int a = 0; a < A; ++a) {
        tpdiv[(size_t)a * (TE + 2)] = 0.0;
        for (int n = 1; n <= TE + 1; ++n) tpdiv[(size_t)a * (TE + 2) + n] = tpv[a] / (double)n; // mcts.py:286: / (count + 1)
    }
    double *d_tab = nullptr;
    MP_TRY(upload_tables(ctx, 7, tab, &d_tab));

    // nodes per tree: an episode expands at most one node (|A| action children) and, in closed loop, creates at most ONE
    // observation node -- the descent stops at a new node, which has no children yet (mcts.py:143-149)
    const long cap = 1 + (long)E * ((long)A + (closed_loop ? 1 : 0));
    // step_strategy "subtree" (open loop): the trees kept by mp_uct_step_tree are re-rooted into the other buffer with room
    // for this plan's expansions; a kept subtree only holds nodes of the last `horizon` plans (see uct_plan_impl)
    const bool cont = ctx->tree.armed && ctx->tree.kind == 4 && ctx->tree.K == 0 && !closed_loop && ctx->tree.n_roots == n_roots &&
                      ctx->tree.A == A && ctx->tree.sp == (pol != nullptr);
    long cap_use = cap;
    if (cont) {
        const long now = 1 + (long)H * E * A;
        if (now > ctx->tree.kept_bound) ctx->tree.kept_bound = now;
        cap_use = (ctx->tree.cap < ctx->tree.kept_bound ? (long)ctx->tree.cap : ctx->tree.kept_bound) + (long)E * A;
    }
    const bool p16 = cap_use <= 65535 && !pol; // (the per-state-policy kernels are built with 32-bit path entries only)
    // path stack entries: the root + one per level, two per level in closed loop (action node, observation node)
    const size_t lds = (ntab + (wbk == 1 ? 256 : 0)) * sizeof(double) + (size_t)(closed_loop ? 2 * H + 2 : H + 2) * 64 * (p16 ? sizeof(uint16_t) : sizeof(int32_t));
    if (lds > 64 * 1024) return fail(MP_ERR_ARG, "mp_uct_plan_stochastic: horizon %d needs %zu B of LDS (> 64 KiB)", H, lds);

    StochArgs a;
    memset(&a, 0, sizeof(a));
    a.n_roots = n_roots; a.mode = mode; a.S = S; a.A = A; a.W = W; a.episodes = E; a.horizon = H; a.cap = (int)cap_use;
    a.closed_loop = closed_loop ? 1 : 0; a.done_on_next = model->done_on_next; a.max_steps = model->max_steps;
    a.table_n = TE;
    a.max_plan_len = max_plan_len;
    a.T = model->T; a.thr = model->thr; a.nxt = model->NXT; a.R = model->R; a.term = model->term; a.tab = d_tab;
    a.srec = wb ? model->srec : nullptr;
    a.rtab = model->srec_rtab;
    a.pol_prior = pol ? pol->prior : nullptr; a.pol_thr = pol ? pol->thr : nullptr; a.pol_mask = pol ? pol->lmask : nullptr;
    a.pol_listed = pol ? pol->listed8 : nullptr;
    a.pol_rslot = pol ? pol->rslot : nullptr; a.pol_stride = pol ? pol->stride : 0; a.temperature = temperature;
    MP_TRY(ws_get(ctx, WS_TREE1, (size_t)n_roots, &a.n_nodes_out)); // per-root tree sizes (updated in place by re-rooting and planning)
    if (cont) {
        MP_TRY(uct_stoch_reroot_now(ctx, cap_use));
        a.hot = (SHot *)ctx->ws[ctx->tree.buf ? WS_TREE5 : WS_TREE0].p;
        a.cold = (SCold *)ctx->ws[ctx->tree.sp && ctx->tree.buf ? WS_TREE6 : WS_TREE2].p;
        a.n_nodes_in = a.n_nodes_out;
    } else {
        ctx->tree.buf = 0;
        ctx->tree.sp = pol != nullptr;
        ctx->tree.kept_bound = 1 + (long)H * E * A;
        MP_TRY(ws_get(ctx, WS_TREE0, (size_t)n_roots * cap_use, &a.hot));
        MP_TRY(ws_get(ctx, WS_TREE2, (size_t)n_roots * cap_use, &a.cold));
    }
    ctx->tree.kind = 4; ctx->tree.n_roots = n_roots; ctx->tree.A = A; ctx->tree.cap = (int)cap_use; ctx->tree.armed = false;
    ctx->tree.K = closed_loop ? 1 : 0;

    int32_t *d_rs = nullptr, *d_st = nullptr;
    MP_TRY(stage_in(ctx, WS_IO0, root_state, (size_t)n_roots, amem, &d_rs));
    if (root_steps) MP_TRY(stage_in(ctx, WS_IO1, root_steps, (size_t)n_roots, amem, &d_st));
    MP_TRY(stage_in(ctx, WS_IO2, (const uint64_t *)rng_state, (size_t)n_roots * 6, rmem, &a.rng));
    uint64_t *d_erng = nullptr;
    if (env_rng_state) MP_TRY(stage_in(ctx, WS_IO9, env_rng_state, (size_t)n_roots * 6, amem, &d_erng));
    a.env_rng = d_erng;
    a.root_state = d_rs; a.root_steps = d_st;
    MP_TRY(stage_out_alloc(ctx, WS_IO3, plans, (size_t)n_roots * max_plan_len, amem, &a.plans));
    MP_TRY(stage_out_alloc(ctx, WS_IO4, plan_len, (size_t)n_roots, amem, &a.plan_len));
    MP_TRY(stage_out_alloc(ctx, WS_IO5, root_value, (size_t)n_roots, amem, &a.root_value));
    MP_TRY(stage_out_alloc(ctx, WS_IO6, root_child_count, (size_t)n_roots * A, amem, &a.root_child_count));
    MP_TRY(stage_out_alloc(ctx, WS_IO7, root_child_value, (size_t)n_roots * A, amem, &a.root_child_value));
    MP_TRY(stage_out_alloc(ctx, WS_IO8, env_steps, (size_t)n_roots, amem, &a.env_steps));

    // mp_uct_record_visits armed this call: every env step of the plan is counted by the state it lands in
    int32_t *visits_host = ctx->visits_host;
    ctx->visits_host = nullptr;
    if (visits_host) {
        if (amem != MP_MEM_HOST) return fail(MP_ERR_ARG, "mp_uct_record_visits: the recording call takes host arrays");
        MP_TRY(ws_get(ctx, WS_VI4, (size_t)n_roots * S, &a.visits)); // (a value-iteration slot: idle during a plan)
        MP_HIP(hipMemsetAsync(a.visits, 0, (size_t)n_roots * S * sizeof(int32_t), st));
    }
    MP_TRY(kernels_begin(ctx));
    {
        typedef void (*kernel_t)(StochArgs);
#define MP_ROW(WBV, PTV) {uct_stoch_kernel<WBV, 0, PTV>, uct_stoch_kernel<WBV, 0, PTV>, uct_stoch_kernel<WBV, 2, PTV>, \
                          uct_stoch_kernel<WBV, 3, PTV>, uct_stoch_kernel<WBV, 4, PTV>, uct_stoch_kernel<WBV, 5, PTV>, \
                          uct_stoch_kernel<WBV, 6, PTV>, uct_stoch_kernel<WBV, 7, PTV>, uct_stoch_kernel<WBV, 8, PTV>}
        static const kernel_t table16[4][9] = {MP_ROW(0, uint16_t), MP_ROW(2, uint16_t), MP_ROW(4, uint16_t), MP_ROW(1, uint16_t)};
        static const kernel_t table32[4][9] = {MP_ROW(0, int32_t), MP_ROW(2, int32_t), MP_ROW(4, int32_t), MP_ROW(1, int32_t)};
#undef MP_ROW
#define MP_ROWP(WBV) {uct_stoch_kernel<WBV, 2, int32_t, true>, uct_stoch_kernel<WBV, 3, int32_t, true>, uct_stoch_kernel<WBV, 4, int32_t, true>, \
                      uct_stoch_kernel<WBV, 5, int32_t, true>, uct_stoch_kernel<WBV, 6, int32_t, true>, uct_stoch_kernel<WBV, 7, int32_t, true>, \
                      uct_stoch_kernel<WBV, 8, int32_t, true>}
        static const kernel_t table_sp[4][7] = {MP_ROWP(0), MP_ROWP(2), MP_ROWP(4), MP_ROWP(1)};
#undef MP_ROWP
        const char *ag = getenv("MP_UCT_STOCH_GENERIC_A"); // "1": the loop form of the selection for any |A| -- test hook
        const int at = A >= 2 && A <= 8 && !(ag && ag[0] == '1') ? A : 0;
        const int wrow = wbk == 2 ? 1 : wbk == 4 ? 2 : wbk == 1 ? 3 : 0;
        if (pol) {
            static const kernel_t any_sp[4] = {uct_stoch_kernel<0, 0, int32_t, true>, uct_stoch_kernel<2, 0, int32_t, true>,
                                               uct_stoch_kernel<4, 0, int32_t, true>, uct_stoch_kernel<1, 0, int32_t, true>};
            if (A < 2) return fail(MP_ERR_ARG, "mp_uct_plan_stochastic_policy: |A| = %d", A);
            // |A| in 2..8: the unrolled forms; beyond (round 4): the loop forms -- any number of actions
            hipLaunchKernelGGL(at ? table_sp[wrow][A - 2] : any_sp[wrow], dim3((unsigned)((n_roots + 63) / 64)), dim3(64), lds, st, a);
        } else {
            hipLaunchKernelGGL((p16 ? table16 : table32)[wrow][at], dim3((unsigned)((n_roots + 63) / 64)), dim3(64), lds, st, a);
        }
    }
    MP_TRY(kernels_end(ctx, 1));
    MP_HIP(hipGetLastError());

    MP_TRY(stage_out_copy(ctx, rng_state, a.rng, (size_t)n_roots * 6, rmem));
    MP_TRY(stage_out_copy(ctx, plans, a.plans, (size_t)n_roots * max_plan_len, amem));
    MP_TRY(stage_out_copy(ctx, plan_len, a.plan_len, (size_t)n_roots, amem));
    MP_TRY(stage_out_copy(ctx, root_value, a.root_value, (size_t)n_roots, amem));
    MP_TRY(stage_out_copy(ctx, root_child_count, a.root_child_count, (size_t)n_roots * A, amem));
    MP_TRY(stage_out_copy(ctx, root_child_value, a.root_child_value, (size_t)n_roots * A, amem));
    MP_TRY(stage_out_copy(ctx, env_steps, a.env_steps, (size_t)n_roots, amem));
    if (visits_host) MP_HIP(hipMemcpyAsync(visits_host, a.visits, (size_t)n_roots * S * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (amem == MP_MEM_HOST) MP_HIP(hipStreamSynchronize(st));
    return MP_OK;
}

extern "C" {

int mp_uct_record_visits(mp_ctx *ctx, int32_t *visits)
{
    if (!ctx) return fail(MP_ERR_ARG, "mp_uct_record_visits: NULL ctx");
    ctx->visits_host = visits; // consumed (and cleared) by the next mp_uct_plan_stochastic / _policy call of this ctx
    return MP_OK;
}

int mp_uct_plan_stochastic(mp_ctx *ctx, mp_model *model, int32_t n_roots, const int32_t *root_state, const int32_t *root_steps,
                           int32_t episodes, int32_t horizon, double gamma, double temperature, const double *prior_p,
                           const double *rollout_p, int32_t closed_loop, uint64_t *rng_state, const uint64_t *env_rng_state,
                           int32_t max_plan_len, int32_t *plans, int32_t *plan_len, double *root_value,
                           int64_t *root_child_count, double *root_child_value, int64_t *env_steps, int32_t mem)
{
    return uct_stoch_plan_impl(ctx, model, nullptr, n_roots, root_state, root_steps, episodes, horizon, gamma, temperature, prior_p,
                               rollout_p, closed_loop, rng_state, env_rng_state, max_plan_len, plans, plan_len, root_value,
                               root_child_count, root_child_value, env_steps, mem);
}

int mp_uct_plan_stochastic_policy(mp_ctx *ctx, mp_model *model, mp_policy *policy, int32_t n_roots, const int32_t *root_state,
                                  const int32_t *root_steps, int32_t episodes, int32_t horizon, double gamma, double temperature,
                                  int32_t closed_loop, uint64_t *rng_state, const uint64_t *env_rng_state, int32_t max_plan_len,
                                  int32_t *plans, int32_t *plan_len, double *root_value, int64_t *root_child_count,
                                  double *root_child_value, int64_t *env_steps, int32_t mem)
{
    if (!policy) return fail(MP_ERR_ARG, "mp_uct_plan_stochastic_policy: policy is NULL");
    return uct_stoch_plan_impl(ctx, model, policy, n_roots, root_state, root_steps, episodes, horizon, gamma, temperature, nullptr,
                               nullptr, closed_loop, rng_state, env_rng_state, max_plan_len, plans, plan_len, root_value,
                               root_child_count, root_child_value, env_steps, mem);
}

int mp_env_step_stochastic(mp_ctx *ctx, mp_model *model, int32_t n, int32_t *state, int32_t *steps, uint8_t *alive,
                           const int32_t *plans, int32_t plan_stride, int32_t max_steps, const double *gpow, double *returns,
                           double *discounted, int32_t *actions_log, int32_t log_stride, int32_t *n_alive, uint64_t *env_rng,
                           int32_t mem)
{
    if (!ctx || !model || !state || !steps || !alive || !plans || !gpow || !returns || !discounted || !n_alive || !env_rng)
        return fail(MP_ERR_ARG, "mp_env_step_stochastic: NULL argument");
    if (mem != MP_MEM_DEVICE) return fail(MP_ERR_ARG, "mp_env_step_stochastic: device arrays only");
    if (model->mode != MP_MODE_STOCHASTIC && model->mode != MP_MODE_SPARSE)
        return fail(MP_ERR_MODE, "mp_env_step_stochastic: stochastic / sparse models only (mp_env_step steps table models)");
    if (model->mode == MP_MODE_STOCHASTIC && (model->M != 1 || model->Sc != model->S))
        return fail(MP_ERR_MODE, "mp_env_step_stochastic: one full dense model [S,A,S] expected");
    if (n < 1 || plan_stride < 1 || max_steps < 1) return fail(MP_ERR_ARG, "mp_env_step_stochastic: bad sizes");
    MP_HIP(hipSetDevice(ctx->device));
    MP_TRY(ensure_thresholds(ctx, model));
    MP_HIP(hipMemsetAsync(n_alive, 0, sizeof(int32_t), ctx->stream));
    const int W = model->mode == MP_MODE_STOCHASTIC ? model->S : model->B;
    hipLaunchKernelGGL(env_step_stoch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, model->A,
                       model->mode == MP_MODE_SPARSE ? 1 : 0, W, model->thr, model->NXT, model->R, model->term, model->done_on_next,
                       state, steps, alive, plans, plan_stride, max_steps, gpow, returns, discounted, actions_log, log_stride, n_alive,
                       env_rng);
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_uct_stoch_tree_priors(mp_ctx *ctx, int32_t cap, double *prior)
{
    if (!ctx || !prior) return fail(MP_ERR_ARG, "mp_uct_stoch_tree_priors: NULL argument");
    if ((size_t)cap < ctx->stoch_priors.size()) return fail(MP_ERR_ARG, "mp_uct_stoch_tree_priors: capacity %d < %zu nodes", cap, ctx->stoch_priors.size());
    for (size_t i = 0; i < ctx->stoch_priors.size(); ++i) prior[i] = ctx->stoch_priors[i];
    return MP_OK;
}

int mp_uct_stoch_tree_capacity(mp_ctx *ctx, int32_t *cap)
{
    if (!ctx || !cap) return fail(MP_ERR_ARG, "mp_uct_stoch_tree_capacity: NULL argument");
    if (ctx->tree.kind != 4) return fail(MP_ERR_ARG, "mp_uct_stoch_tree_capacity: no such tree on this ctx");
    *cap = ctx->tree.cap;
    return MP_OK;
}

int mp_uct_stoch_tree_export(mp_ctx *ctx, int32_t root, int32_t cap, int32_t *n_nodes, int32_t *parent, int32_t *key,
                             uint8_t *is_obs, int64_t *count, double *value)
{
    if (!ctx) return fail(MP_ERR_ARG, "ctx is NULL");
    if (ctx->tree.kind != 4) return fail(MP_ERR_ARG, "mp_uct_stoch_tree_export: no tree of mp_uct_plan_stochastic on this ctx");
    if (root < 0 || root >= ctx->tree.n_roots) return fail(MP_ERR_ARG, "mp_uct_stoch_tree_export: root %d out of range", root);
    MP_HIP(hipSetDevice(ctx->device));
    MP_HIP(hipStreamSynchronize(ctx->stream));
    int32_t n = 0;
    MP_HIP(hipMemcpy(&n, (const int32_t *)ctx->ws[WS_TREE1].p + root, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (n > cap) return fail(MP_ERR_ARG, "mp_uct_stoch_tree_export: capacity %d < %d nodes", cap, n);
    std::vector<SHot> h((size_t)n);
    std::vector<SCold> c((size_t)n);
    MP_HIP(hipMemcpy(h.data(), (const SHot *)ctx->ws[ctx->tree.buf ? WS_TREE5 : WS_TREE0].p + (long)root * ctx->tree.cap, (size_t)n * sizeof(SHot),
                     hipMemcpyDeviceToHost));
    MP_HIP(hipMemcpy(c.data(), (const SCold *)ctx->ws[ctx->tree.sp && ctx->tree.buf ? WS_TREE6 : WS_TREE2].p + (long)root * ctx->tree.cap,
                     (size_t)n * sizeof(SCold), hipMemcpyDeviceToHost));
    // node types by one pass in creation order (a parent is older than its children): the root and the observation nodes
    // carry a cold half; an action node's key / parent follow from the sibling group it sits in
    const int A = ctx->tree.A;
    const bool closed = ctx->tree.K != 0;
    std::vector<int32_t> par((size_t)n, -1), ky((size_t)n, -1);
    std::vector<uint8_t> obs((size_t)n, 0), act((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        const int f = h[i].first;
        if (f < 0) continue;
        if (act[i] && closed) { // an action node of a closed-loop tree: its observation children, linked by `next`
            for (int o = f; o >= 0 && o < n; o = c[o].next) { obs[o] = 1; par[o] = i; ky[o] = c[o].key; }
        } else {                // root / observation node (open loop: any node): |A| contiguous action children
            for (int a = 0; a < A && f + a < n; ++a) { act[f + a] = 1; par[f + a] = i; ky[f + a] = a; }
        }
    }
    // per-state policies: the slots of unlisted actions are phantoms (count = -1), not nodes: dropped and the ids closed up
    std::vector<int32_t> new_id((size_t)n, -1);
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (!(act[i] && h[i].count < 0)) new_id[i] = m++;
    for (int i = 0; i < n; ++i) {
        const int j = new_id[i];
        if (j < 0) continue;
        if (parent) parent[j] = par[i] >= 0 ? new_id[par[i]] : -1;
        if (key) key[j] = ky[i];
        if (is_obs) is_obs[j] = obs[i];
        if (count) count[j] = h[i].count;
        if (value) value[j] = h[i].value;
    }
    // stored priors of the action nodes (per-state policies), for the agent-level tree: kept for mp_uct_stoch_tree_priors
    ctx->stoch_priors.assign((size_t)m, 0.0);
    if (ctx->tree.sp)
        for (int i = 0; i < n; ++i)
            if (new_id[i] >= 0 && act[i]) memcpy(&ctx->stoch_priors[(size_t)new_id[i]], &c[i], sizeof(double));
    if (n_nodes) *n_nodes = m;
    return MP_OK;
}

} // extern "C"
