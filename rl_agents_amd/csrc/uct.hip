// uct.hip -- MCTS/UCT planning (tree_search/mcts.py) for thousands of independent roots.
//
// Mapping: ONE ROOT PER LANE.  A root's plan() is inherently sequential (every episode reads
// the tree the previous one updated; every env step depends on the previous state), so the
// parallel axis is roots: 64 roots per wavefront, one wavefront per workgroup so that a batch of
// n_roots roots spreads over n_roots/64 SIMDs.  A cloned environment (reference:
// safe_deepcopy_env, common/factory.py:119-134) is an (int32 state, int32 steps) register pair.
//
// Memory layout (HBM, all L2-resident at BASELINE sizes):
//   model   Rec[S*A]  one 16-byte record per (s,a): {next, flags, reward} -> one dwordx4 gather
//           per env step instead of three dependent-latency gathers from T / R / terminal.
//   tree    Node[n_roots][cap]  root-major 16-byte records {value f64, count i32, first_child i32};
//           the A children of a node are contiguous (80 B at A = 5: one or two cache lines per
//           selection level).  cap = 1 + episodes*A (at most one expansion per episode).
//   path    per-lane stack of visited node ids in LDS ([depth][lane], conflict-free), so the
//           backup is a pipelined read-modify-write over known addresses instead of a dependent
//           parent-pointer chase.
// Arithmetic is the reference's, operation for operation, in IEEE double without contraction;
// randomness is numpy's PCG64 stepped on the device (pcg64.hpp), so results are bit-identical
// to the Python planner for equal seeds.
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "common.hpp"
#include "pcg64.hpp"

namespace mp {

struct alignas(16) UctNode {
    double value;
    int32_t count;
    int32_t first_child; // -1 = not expanded
};
static_assert(sizeof(UctNode) == 16, "UctNode must be one dwordx4");

struct UctArgs {
    int n_roots, S, A, episodes, horizon, cap;
    int done_on_next, max_steps, max_plan_len;
    int lanes; // roots per wavefront (64 = dense; fewer spreads a small batch over more SIMDs)
    const Rec *rec;
    const int32_t *root_state, *root_steps;
    const double *gpow; // gamma ** h, h = 0..horizon   (host libm pow, = Python's float **)
    const double *tp;   // temperature * A * prior[a]
    const double *cdf;  // cumsum(rollout_p) / cumsum(rollout_p)[-1]
    uint64_t *rng;
    UctNode *tree;
    int32_t *plans, *plan_len;
    double *root_value, *root_child_value;
    int64_t *root_child_count, *env_steps;
};

// AT > 0: |A| known at compile time (children scored from registers in one pass, tables in
// registers); AT == 0: any |A| (three passes over the children, tables in LDS).
template <int AT>
__global__ __launch_bounds__(64) void uct_table_kernel(UctArgs p)
{
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    const int lane = threadIdx.x;
    const int A = AT > 0 ? AT : p.A, H = p.horizon;
    // LDS: gamma powers [H + 1] | (generic) tp [A], cdf [A] | path stack [(H + 1)][64] int32
    double *gpow = lds_d;
    double *tpL = gpow + (H + 1);
    double *cdfL = tpL + (AT > 0 ? 0 : A);
    int32_t *path = reinterpret_cast<int32_t *>(cdfL + (AT > 0 ? 0 : A));
    for (int i = lane; i <= H; i += 64) gpow[i] = p.gpow[i];
    if (AT == 0)
        for (int i = lane; i < A; i += 64) { tpL[i] = p.tp[i]; cdfL[i] = p.cdf[i]; }
    constexpr int AR = AT > 0 ? AT : 1;
    double tp[AR], cdf[AR];
    if (AT > 0) {
#pragma unroll
        for (int a = 0; a < AR; ++a) { tp[a] = p.tp[a]; cdf[a] = p.cdf[a]; }
    }
    __syncthreads();
    const int r = blockIdx.x * p.lanes + lane;
    if (lane >= p.lanes || r >= p.n_roots) return;
    UctNode *tree = p.tree + (long)r * p.cap;
    const Rec *__restrict__ rec = p.rec;
    Pcg64 g;
    g.load(p.rng + (long)r * 6);
    const int32_t s0 = p.root_state[r];
    const int32_t st0 = p.root_steps ? p.root_steps[r] : 0;
    // mcts.py:129-130 reset(): fresh root
    {
        UctNode n;
        n.value = 0.0; n.count = 0; n.first_child = -1;
        tree[0] = n;
    }
    int n_nodes = 1;
    int64_t steps_taken = 0;
    const uint32_t done_bit = p.done_on_next ? 2u : 1u;

#ifdef MP_PROFILE
    long long t_sel = 0, t_expd = 0, t_roll = 0, t_bak = 0, n_sel = 0, n_roll = 0;
    const long long t_all0 = clock64();
#define PROF_T(x) const long long x = clock64()
#else
#define PROF_T(x)
#endif
    for (int ep = 0; ep < p.episodes; ++ep) { // mcts.py:179-184
        PROF_T(c0);
        int32_t s = s0, st = st0;
        int node = 0, depth = 0;
        bool terminal = false;
        double total = 0.0;
        path[lane] = 0;
        int fc = tree[0].first_child;
        // ---- selection, mcts.py:143-149
        while (depth < H && fc >= 0 && !terminal) {
            // MCTSNode.selection_strategy (mcts.py:275-286) for each child, Node.random_argmax
            // (abstract.py:296-311): exact-equality argmax set, uniform draw among >= 2 ties
            int act = 0;
            if (AT > 0) {
                UctNode c[AR];
#pragma unroll
                for (int a = 0; a < AR; ++a) c[a] = tree[fc + a];
                double sc[AR];
#pragma unroll
                for (int a = 0; a < AR; ++a) sc[a] = c[a].value + tp[a] / (double)(c[a].count + 1);
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
                    if (eq && !found && pick == 0) { act = a; found = true; }
                    if (eq && !found) --pick;
                }
            } else {
                double m = 0.0;
                for (int a = 0; a < A; ++a) {
                    const UctNode c = tree[fc + a];
                    const double sc = c.value + tpL[a] / (double)(c.count + 1);
                    if (a == 0 || sc > m) m = sc;
                }
                int nt = 0;
                for (int a = 0; a < A; ++a) {
                    const UctNode c = tree[fc + a];
                    const double sc = c.value + tpL[a] / (double)(c.count + 1);
                    nt += sc == m ? 1 : 0;
                }
                int pick = (int)g.below((uint32_t)nt);
                for (int a = 0; a < A; ++a) {
                    const UctNode c = tree[fc + a];
                    const double sc = c.value + tpL[a] / (double)(c.count + 1);
                    if (sc == m) {
                        if (pick == 0) { act = a; break; }
                        --pick;
                    }
                }
            }
            const Rec rc = rec[(long)s * A + act];
            terminal = (rc.flags & done_bit) != 0;
            s = rc.next;
            ++st;
            ++steps_taken;
            total += gpow[depth] * rc.reward;
            node = fc + act;
            ++depth;
            path[depth * 64 + lane] = node;
            fc = tree[node].first_child;
#ifdef MP_PROFILE
            ++n_sel;
#endif
        }
        PROF_T(c1);
        // ---- expansion, mcts.py:151-154 / 237-246
        if (fc < 0 && depth < H && (!terminal || node == 0)) {
            tree[node].first_child = n_nodes;
            UctNode n;
            n.value = 0.0; n.count = 0; n.first_child = -1;
            for (int a = 0; a < A; ++a) tree[n_nodes + a] = n;
            n_nodes += A;
        }
        PROF_T(c2);
        // ---- rollout, mcts.py:156-157 / 160-177
        if (!terminal) {
            for (int h = depth; h < H; ++h) {
                const double u = g.next_double();
                // searchsorted(cdf, u, side='right') on a non-decreasing cdf = #{a : cdf[a] <= u}
                int act = 0;
                if (AT > 0) {
#pragma unroll
                    for (int a = 0; a < AR; ++a) act += cdf[a] <= u ? 1 : 0;
                } else {
                    for (int a = 0; a < A; ++a) act += cdfL[a] <= u ? 1 : 0;
                }
                const double gh = gpow[h];
                const Rec rc = rec[(long)s * A + act];
                const bool term_h = (rc.flags & done_bit) != 0;
                s = rc.next;
                ++st;
                ++steps_taken;
                total += gh * rc.reward;
                const bool trunc_h = p.max_steps > 0 && st >= p.max_steps;
#ifdef MP_PROFILE
                ++n_roll;
#endif
                if (term_h || trunc_h) break;
            }
        }
        PROF_T(c3);
        // ---- backup, mcts.py:248-265: the same return for every node on the path
        for (int d = depth; d >= 0; --d) {
            const int n = path[d * 64 + lane];
            UctNode c = tree[n];
            c.count += 1;
            c.value += 1.0 / (double)c.count * (total - c.value);
            tree[n] = c;
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
    g.store(p.rng + (long)r * 6);
    // ---- AbstractPlanner.get_plan (abstract.py:143-156) with MCTSNode.selection_rule
    // (mcts.py:212-218): most visited child, ties -> first maximal value among them
    {
        int node = 0, len = 0;
        int fc = tree[0].first_child;
        while (fc >= 0) {
            int mc = tree[fc].count;
            for (int a = 1; a < A; ++a) mc = max(mc, tree[fc + a].count);
            int best = -1;
            double bv = 0.0;
            for (int a = 0; a < A; ++a) {
                const UctNode c = tree[fc + a];
                if (c.count == mc && (best < 0 || c.value > bv)) { best = a; bv = c.value; }
            }
            if (p.plans && len < p.max_plan_len) p.plans[(long)r * p.max_plan_len + len] = best;
            ++len;
            node = fc + best;
            fc = tree[node].first_child;
        }
        if (p.plans)
            for (int i = len; i < p.max_plan_len; ++i) p.plans[(long)r * p.max_plan_len + i] = -1;
        if (p.plan_len) p.plan_len[r] = len;
    }
    if (p.root_value) p.root_value[r] = tree[0].value;
    if (p.env_steps) p.env_steps[r] = steps_taken;
    const int rfc = tree[0].first_child;
    for (int a = 0; a < A; ++a) {
        if (p.root_child_count) p.root_child_count[(long)r * A + a] = rfc >= 0 ? tree[rfc + a].count : 0;
        if (p.root_child_value) p.root_child_value[(long)r * A + a] = rfc >= 0 ? tree[rfc + a].value : 0.0;
    }
}

// Roots per wavefront.  A root's episodes are one long dependency chain, so a batch takes as long
// as its slowest wavefront; with few roots it pays to spread them thin (fewer lanes per wave = less
// rollout-length divergence and fewer distinct cache lines per gather) until every SIMD has a wave.
static int uct_lanes_per_wave(const mp_ctx *ctx, int n_roots)
{
    if (const char *e = getenv("MP_UCT_LANES")) {
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) return v;
    }
    const int simds = ctx->prop.multiProcessorCount * 4;
    int lanes = 64;
    while (lanes > 4 && (long)n_roots < (long)simds * lanes) lanes >>= 1;
    return lanes;
}

template <int AT>
static void uct_launch(const UctArgs &a, size_t lds, hipStream_t st)
{
    hipLaunchKernelGGL(uct_table_kernel<AT>, dim3((unsigned)((a.n_roots + a.lanes - 1) / a.lanes)), dim3(64), lds, st, a);
}

} // namespace mp

using namespace mp;

extern "C" {

int mp_uct_plan(mp_ctx *ctx, mp_model *model, int32_t n_roots, const void *root_state, const int32_t *root_steps,
                int32_t episodes, int32_t horizon, double gamma, double temperature, const double *prior_p,
                const double *rollout_p, uint64_t *rng_state, int32_t max_plan_len, int32_t *plans,
                int32_t *plan_len, double *root_value, int64_t *root_child_count, double *root_child_value,
                int64_t *env_steps, int32_t mem)
{
    if (!ctx || !model || !root_state || !prior_p || !rollout_p || !rng_state)
        return fail(MP_ERR_ARG, "mp_uct_plan: NULL argument");
    if (model->mode != MP_MODE_DETERMINISTIC)
        return fail(MP_ERR_MODE, "mp_uct_plan: model mode %d is not a deterministic table", model->mode);
    if (n_roots < 1 || episodes < 0 || horizon < 0 || max_plan_len < 0)
        return fail(MP_ERR_ARG, "mp_uct_plan: bad sizes (n_roots=%d episodes=%d horizon=%d)", n_roots, episodes, horizon);
    const int A = model->A;
    const size_t lds = (size_t)(horizon + 1) * (64 * sizeof(int32_t) + sizeof(double)) + 2 * (size_t)A * sizeof(double);
    if (lds > 64 * 1024) return fail(MP_ERR_ARG, "mp_uct_plan: horizon %d too deep for the LDS path stack", horizon);
    MP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const long cap = 1 + (long)episodes * A;

    // small per-call tables, computed on the host exactly as Python computes them
    std::vector<double> tab((size_t)horizon + 1 + 2 * (size_t)A);
    for (int h = 0; h <= horizon; ++h) tab[h] = pow(gamma, (double)h);           // gamma ** h
    double *tp = tab.data() + horizon + 1, *cdf = tp + A;
    for (int a = 0; a < A; ++a) tp[a] = temperature * (double)A * prior_p[a];      // mcts.py:286, left to right
    double acc = 0.0;
    for (int a = 0; a < A; ++a) { acc += rollout_p[a]; cdf[a] = acc; }             // numpy cumsum
    for (int a = 0; a < A; ++a) cdf[a] /= acc;                                     // cdf /= cdf[-1]
    double *d_tab = nullptr;
    MP_TRY(upload_tables(ctx, 1, tab, &d_tab));

    UctArgs a;
    a.n_roots = n_roots; a.S = model->S; a.A = A; a.episodes = episodes; a.horizon = horizon; a.cap = (int)cap;
    a.done_on_next = model->done_on_next; a.max_steps = model->max_steps; a.max_plan_len = max_plan_len;
    a.lanes = uct_lanes_per_wave(ctx, n_roots);
    a.rec = model->rec;
    a.gpow = d_tab; a.tp = d_tab + horizon + 1; a.cdf = a.tp + A;
    MP_TRY(ws_get(ctx, WS_TREE0, (size_t)n_roots * cap, &a.tree));
    ctx->tree.kind = 1; ctx->tree.n_roots = n_roots; ctx->tree.A = A; ctx->tree.cap = (int)cap;

    int32_t *d_rs = nullptr, *d_st = nullptr;
    MP_TRY(stage_in(ctx, WS_IO0, (const int32_t *)root_state, (size_t)n_roots, mem, &d_rs));
    if (root_steps) MP_TRY(stage_in(ctx, WS_IO1, root_steps, (size_t)n_roots, mem, &d_st));
    MP_TRY(stage_in(ctx, WS_IO2, (const uint64_t *)rng_state, (size_t)n_roots * 6, mem, &a.rng));
    a.root_state = d_rs; a.root_steps = d_st;
    MP_TRY(stage_out_alloc(ctx, WS_IO3, plans, (size_t)n_roots * max_plan_len, mem, &a.plans));
    MP_TRY(stage_out_alloc(ctx, WS_IO4, plan_len, (size_t)n_roots, mem, &a.plan_len));
    MP_TRY(stage_out_alloc(ctx, WS_IO5, root_value, (size_t)n_roots, mem, &a.root_value));
    MP_TRY(stage_out_alloc(ctx, WS_IO6, root_child_count, (size_t)n_roots * A, mem, &a.root_child_count));
    MP_TRY(stage_out_alloc(ctx, WS_IO7, root_child_value, (size_t)n_roots * A, mem, &a.root_child_value));
    MP_TRY(stage_out_alloc(ctx, WS_IO8, env_steps, (size_t)n_roots, mem, &a.env_steps));

    MP_TRY(kernels_begin(ctx));
    switch (A) {
    case 2: uct_launch<2>(a, lds, st); break;
    case 3: uct_launch<3>(a, lds, st); break;
    case 4: uct_launch<4>(a, lds, st); break;
    case 5: uct_launch<5>(a, lds, st); break;
    case 6: uct_launch<6>(a, lds, st); break;
    case 8: uct_launch<8>(a, lds, st); break;
    default: uct_launch<0>(a, lds, st); break;
    }
    MP_TRY(kernels_end(ctx, 1));
    MP_HIP(hipGetLastError());

    MP_TRY(stage_out_copy(ctx, rng_state, a.rng, (size_t)n_roots * 6, mem));
    MP_TRY(stage_out_copy(ctx, plans, a.plans, (size_t)n_roots * max_plan_len, mem));
    MP_TRY(stage_out_copy(ctx, plan_len, a.plan_len, (size_t)n_roots, mem));
    MP_TRY(stage_out_copy(ctx, root_value, a.root_value, (size_t)n_roots, mem));
    MP_TRY(stage_out_copy(ctx, root_child_count, a.root_child_count, (size_t)n_roots * A, mem));
    MP_TRY(stage_out_copy(ctx, root_child_value, a.root_child_value, (size_t)n_roots * A, mem));
    MP_TRY(stage_out_copy(ctx, env_steps, a.env_steps, (size_t)n_roots, mem));
    if (mem == MP_MEM_HOST) MP_HIP(hipStreamSynchronize(st));
    return MP_OK;
}

int mp_uct_tree_export(mp_ctx *ctx, int32_t root, int32_t cap, int32_t *n_nodes, int32_t *parent, int32_t *action,
                       int64_t *count, double *value, int32_t *first_child)
{
    if (!ctx) return fail(MP_ERR_ARG, "ctx is NULL");
    if (ctx->tree.kind != 1) return fail(MP_ERR_ARG, "mp_uct_tree_export: no UCT tree on this ctx");
    if (root < 0 || root >= ctx->tree.n_roots) return fail(MP_ERR_ARG, "mp_uct_tree_export: root %d out of range", root);
    const int tcap = ctx->tree.cap, A = ctx->tree.A;
    std::vector<UctNode> h((size_t)tcap);
    MP_HIP(hipSetDevice(ctx->device));
    MP_HIP(hipStreamSynchronize(ctx->stream));
    MP_HIP(hipMemcpy(h.data(), (const UctNode *)ctx->ws[WS_TREE0].p + (long)root * tcap, (size_t)tcap * sizeof(UctNode),
                     hipMemcpyDeviceToHost));
    // nodes are appended A at a time; the tree in use is the closure of first_child links
    int n = 1;
    for (int i = 0; i < n && i < tcap; ++i)
        if (h[i].first_child >= 0 && h[i].first_child + A > n) n = h[i].first_child + A;
    if (n > cap) return fail(MP_ERR_ARG, "mp_uct_tree_export: capacity %d < %d nodes", cap, n);
    if (parent) parent[0] = -1;
    if (action) action[0] = -1;
    for (int i = 0; i < n; ++i) {
        if (count) count[i] = h[i].count;
        if (value) value[i] = h[i].value;
        if (first_child) first_child[i] = h[i].first_child;
        if (h[i].first_child >= 0)
            for (int a = 0; a < A; ++a) {
                if (parent) parent[h[i].first_child + a] = i;
                if (action) action[h[i].first_child + a] = a;
            }
    }
    if (n_nodes) *n_nodes = n;
    return MP_OK;
}

} // extern "C"
