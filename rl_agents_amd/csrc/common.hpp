// common.hpp -- context, model and error plumbing shared by the kernels of libmi355plan.so.
// gfx950 only: wave = 64 lanes, 160 KiB LDS per CU, 256 CUs in 8 XCDs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mi355plan.h"

namespace mp {

constexpr int kWave = 64;
constexpr size_t kLdsBytes = 160 * 1024;

extern thread_local std::string g_err;

int fail(int code, const char *fmt, ...);

#define MP_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return mp::fail(MP_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),    \
                            __FILE__, __LINE__);                                                  \
    } while (0)

#define MP_TRY(call)                  \
    do {                              \
        int rc_ = (call);             \
        if (rc_ != MP_OK) return rc_; \
    } while (0)

// One packed transition record of a deterministic table model: everything an env step needs in
// a single 16-byte gather (replaces the reference's env.step on a deep-copied env object).
struct alignas(16) Rec {
    int32_t next;   // transition[s, a]
    uint32_t flags; // bit0 = terminal[s] (state acted from), bit1 = terminal[next], bit2 = action available in s
                    // (fused policy records, mp_policy: bits 8-15 = actions the prior policy lists in `next`, 16-23 = in s)
    double reward;  // reward[s, a]
};
static_assert(sizeof(Rec) == 16, "Rec must be one dwordx4");

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

// what the last tree-search call left on the device (for the *_tree_export entry points)
struct TreeMeta {
    int kind = 0; // 0 none, 1 uct, 2 opd, 3 robust opd
    int n_roots = 0, A = 0, cap = 0, K = 0, M = 1;
    double gamma = 0.0; // robust opd: the export recomputes leaf upper-bound vectors
    int buf = 0;        // UCT: which of the two tree workspaces (WS_TREE0 / WS_TREE2) holds the current trees
    bool armed = false; // UCT: mp_uct_step_tree was called; the next plan re-roots and continues
    int il = 0;          // UCT: 1 = wave-interleaved tree layout (see uct.hip TreeRef)
    long kept_bound = 0; // UCT: upper bound on the nodes a kept (re-rooted) tree can hold, see uct_plan_impl
    bool sp = false;     // stochastic UCT: the last plan used per-state policies (stored priors in the cold halves, phantom slots)
};

enum { WS_TREE0 = 0, WS_TREE1, WS_TREE2, WS_TREE3, WS_TREE4, WS_TREE5, WS_TREE6, WS_TREE7, WS_IO0, WS_IO1, WS_IO2, WS_IO3, WS_IO4,
       WS_IO5, WS_IO6, WS_IO7, WS_IO8, WS_IO9, WS_TAB0, WS_TAB1, WS_TAB2, WS_TAB3, WS_VI0, WS_VI1, WS_VI2, WS_VI3,
       WS_VI4, WS_VI5, WS_GROOT, WS_JUMP, WS_POL0, WS_POL1, WS_POL2, WS_POL3, WS_POL4, WS_COUNT };

// everything a captured chain of deterministic VI sweeps bakes into its kernel arguments
struct ViGraphKey {
    const void *model, *T, *R, *term, *Vb, *notclose;
    int iterations, M, S, A, robust, vform;
    double gamma, rtol, atol;
};

} // namespace mp

struct mp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    int n_launches = 0;
    char last_variant[48] = ""; // which kernel variant the last tree-search call launched (mp_last_kernel_variant)
    hipDeviceProp_t prop;
    mp::DevBuf ws[mp::WS_COUNT];
    mp::TreeMeta tree;
    // host mirror of the small per-call tables (gamma powers, priors, cdf) last uploaded to
    // WS_TAB0: identical parameters on the next call skip the upload and its stream sync
    std::vector<double> tab_host;
    int tab_kind = 0;
    // cached hipGraphExec of the last deterministic VI sweep chain
    void *vi_graph_exec = nullptr;
    mp::ViGraphKey vi_graph_key;
    // dense value iteration: 1 = contract in numpy's summation order (vi_dense_exact_q, bit-exact with the reference),
    // 0 = on the f64 matrix cores (vi_dense_q); -1 = not set yet (MP_VI_DENSE in the environment, else the default)
    int vi_dense_exact = -1;
    int vi_exact_cols = 0; // the row length whose summation plan sits in WS_VI5 (0 = none)
    int vi_exact_nleaf = 0, vi_exact_nnode = 0, vi_exact_nh = 0, vi_exact_nb = 0, vi_exact_npiece = 0;
    // side streams of the pipelined host-mode plan (created on first use), one completion event each, one fork event
    hipStream_t pipe[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t pipe_done[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t pipe_fork = nullptr;
    // host blocks handed out by mp_host_alloc: pinned AND mapped into the device's address space, so a kernel can read
    // root states from / write results to them directly (zero-copy: no hipMemcpy call at all on the host-array path)
    struct Pinned { const char *host; char *dev; size_t bytes; };
    std::vector<Pinned> pinned;
    // Device blocks of planner objects that come and go on this ctx (mp_saopd_create / _free): a freed block is kept -- at
    // most kBlockCacheBytes of them -- and handed to the next allocation of the same size.  hipFree synchronises the device
    // and costs ~0.2 ms a call; a batch of state-aware planners is twenty blocks.  Every user of the blocks works on the
    // ctx's stream, so a block that is handed out again is reused in stream order.
    struct Block { void *p; size_t bytes; };
    std::vector<Block> block_cache;
    void *comm = nullptr;             // ncclComm_t of mp_comm_init (RCCL resolved with dlopen: api.hip)
    int comm_world = 0;
    int jump_entries = 0;             // PCG64 jump-ahead table (limbs of A^n, G_n for n < jump_entries) resident in WS_JUMP
    std::vector<std::vector<uint32_t>> jump_host; // its host copies (one per rebuild): the source of an ASYNCHRONOUS upload has to outlive the call
    int32_t *visits_host = nullptr;   // mp_uct_record_visits: where the next stochastic-kernel plan writes its visit counts
    std::vector<double> stoch_priors; // stored child priors of the tree last exported by mp_uct_stoch_tree_export (per-state policies)
    size_t block_cache_bytes = 0;
    // Sticky fault word for ASYNCHRONOUS device-array calls (pinned host memory mapped into the device: kernels add to it
    // through fault_dev, the host reads fault_host after a synchronisation).  mp_uct_plan_models / mp_opd_plan_models on
    // device arrays cannot validate model_index / root_state on the host: the globalize kernel clamps a bad root to state 0
    // of model 0 and counts it here; mp_ctx_synchronize reports and clears it (mp_ctx_device_faults reads it without clearing).
    int32_t *fault_host = nullptr, *fault_dev = nullptr;
};

constexpr size_t kBlockCacheBytes = (size_t)2 << 30;

// give every cached block back to the runtime (an allocation failed: dead blocks of other sizes may be what is in the way)
inline void ctx_block_cache_flush(mp_ctx *ctx)
{
    for (auto &b : ctx->block_cache) (void)hipFree(b.p);
    ctx->block_cache.clear();
    ctx->block_cache_bytes = 0;
}

// hipMalloc through the ctx's block cache: the smallest cached block that holds `bytes` and wastes at most a quarter of
// itself (a batch of planners re-created with slightly different sizes reuses blocks instead of piling up dead ones);
// when the runtime is out of memory the cache is flushed and the allocation tried once more
inline hipError_t ctx_block_alloc(mp_ctx *ctx, void **out, size_t bytes, size_t *got = nullptr)
{
    if (got) *got = bytes;
    size_t best = ctx->block_cache.size();
    for (size_t i = 0; i < ctx->block_cache.size(); ++i) {
        const size_t have = ctx->block_cache[i].bytes;
        if (have >= bytes && have - bytes <= have / 4 && (best == ctx->block_cache.size() || have < ctx->block_cache[best].bytes)) best = i;
    }
    if (best != ctx->block_cache.size()) {
        *out = ctx->block_cache[best].p;
        if (got) *got = ctx->block_cache[best].bytes;    // (the block's real size: what goes back into the cache later)
        ctx->block_cache_bytes -= ctx->block_cache[best].bytes;
        ctx->block_cache[best] = ctx->block_cache.back();
        ctx->block_cache.pop_back();
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess && !ctx->block_cache.empty()) {
        (void)hipGetLastError();
        ctx_block_cache_flush(ctx);
        e = hipMalloc(out, bytes);
    }
    return e;
}
template <typename T>
inline hipError_t ctx_block_alloc(mp_ctx *ctx, T **out, size_t bytes, size_t *got = nullptr) { return ctx_block_alloc(ctx, reinterpret_cast<void **>(out), bytes, got); }

// is this ctx still alive?  (api.hip keeps the set of live contexts: an object may outlive the ctx it was made on)
bool mp_ctx_alive(const mp_ctx *ctx);

// back into the cache, or hipFree when the cache is full or the ctx is gone
inline void ctx_block_release(mp_ctx *ctx, void *p, size_t bytes)
{
    if (!p) return;
    if (ctx && bytes && mp_ctx_alive(ctx) && ctx->block_cache_bytes + bytes <= kBlockCacheBytes && ctx->block_cache.size() < 256) {
        ctx->block_cache.push_back({p, bytes});
        ctx->block_cache_bytes += bytes;
    } else {
        (void)hipFree(p);
    }
}

// device-resident numpy-PCG64 generator records of a batch of roots
struct mp_rng {
    mp_ctx *ctx = nullptr;
    int n = 0;
    uint64_t *state = nullptr; // [n][6]
};

namespace mp {
inline uint64_t next_model_serial()
{
    static uint64_t serial = 0; // models are created under the caller's serialisation of a ctx (see mi355plan.h)
    return ++serial;
}
} // namespace mp

struct mp_model {
    mp_ctx *ctx = nullptr;
    uint64_t serial = mp::next_model_serial(); // distinguishes a model from a later one at the same address
    int mode = 0, M = 1, S = 0, A = 0, B = 0;
    int Sc = 0; // dense models: number of next-state columns (= S, or the full |S| for a block of rows)
    // batch models (mp_model_load_table_batch): NB independent MDPs of Sb states each, S = NB * Sb GLOBAL states
    // (b * Sb + s); every table below is the disjoint union.  NB = 1, Sb = S for any other model.
    int NB = 1, Sb = 0;
    // table updates (mp_model_update_tables / _rows): a pinned staging block + a device scratch block owned by the model,
    // and the event that says the last update's copies out of the staging block are done
    void *upd_stage = nullptr;
    size_t upd_stage_cap = 0;
    void *upd_dev = nullptr;
    size_t upd_dev_cap = 0;
    hipEvent_t upd_done = nullptr;
    bool upd_pending = false;
    std::unordered_map<uint64_t, int> *rmap = nullptr; // host mirror of rdict (bit pattern -> index): updates extend it
    std::vector<void *> dead_blocks;                   // device arrays an update retired (freed with the model)
    int done_on_next = 0, max_steps = 0;
    bool masked = false;     // mp_model_set_available restricted the action sets (state.get_available_actions())
    uint8_t *avail = nullptr; // device [S*A] flags, only when masked
    // deterministic tables, device, [M,S,A]
    int32_t *T = nullptr;
    double *R = nullptr;
    uint8_t *term = nullptr; // [S] or nullptr
    mp::Rec *rec = nullptr;  // packed model 0, [S*A]
    mp::Rec *rec_all = nullptr;  // joint models (mp_model_load_joint): packed records of every model, [M][S*A]; rec = rec_all
    uint8_t *term_all = nullptr; // joint models: terminal flags per model, [M][S]
    uint16_t *t16 = nullptr; // model 0 as {bit15 = terminal[next], next}, [S*A]; only when S < 32768
    uint8_t *r8 = nullptr;   // model 0's rewards as indices into rdict, [S*A] (padded to 16 B); only with t16 and <= 256 distinct rewards
    double *rdict = nullptr; // the distinct reward values (bit patterns), [256]
    int n_rdict = 0;
    // state-aware OPD (saopd.hip): Bellman backups a fresh planner's first plan ran, by root state -- learned from the
    // batches planned on this model so far and used ONLY to dispatch the planners of the next fresh batch longest first
    int32_t *sa_cost = nullptr; // device [S], lazily allocated, 0 = nothing seen yet
    // dense [M,S,A,S] / sparse [S,A,B]
    const double *P = nullptr;
    bool borrowed = false;
    int32_t *NXT = nullptr;
    uint64_t *thr = nullptr; // dense / sparse models: sampling thresholds ceil(cdf * 2^53) of every row (uct_stoch.hip), lazily built
    uint4 *srec = nullptr;   // sparse models with B <= 4: one fused record per (s, a) -- thresholds, next states, reward, terminal
                             // flags (uct_stoch.hip: an env step is ONE gather), lazily built; 2 (B <= 2) or 4 uint4 each
    int srec_wb = 0;         // 0 = not looked at yet, 2 / 4 = uint4 per record, -1 = rows too wide (dense rows by binary search),
                             // 1 = ONE uint4 per record: two successors at most and at most 256 distinct rewards (srec_rtab)
    double *srec_rtab = nullptr; // srec_wb == 1: the distinct reward values [256] a record's 8-bit index points into
    mp_cartpole_params cp;
    int cp_sincos = 0;       // CartPole: which restated form of the host libm's sin / cos the kernel evaluates (libm_sincos.hpp)
};

// per-state prior / rollout policies of one model (mcts_with_prior.py:47-62), device
struct mp_policy {
    mp_ctx *ctx = nullptr;
    const mp_model *model = nullptr; // the model whose records are fused into frec
    uint64_t model_serial = 0;
    int S = 0, A = 0;
    int stride = 0;             // doubles per row of prior / thr: A rounded up to even (16-byte rows)
    int frq = 0;                // 16-byte chunks per fused record: 1 + ceil((A-1)/4)
    int shift = 21;             // fused thresholds keep thr >> shift (saturated to 32 bits; 43 and 10 bits when packed)
    int listed = 0;             // 1: the prior policy lists a subset of the actions per state (masks in the fused records)
    int packed = 0;             // 1: frec16 holds 16-byte records {next:20|th0:10, flags:2|th1:10|th2:10|th3:10, reward}
    uint4 *frec16 = nullptr;    // [S*A]
    double *prior = nullptr;    // [S][stride]  prior[s][a]
    uint64_t *thr = nullptr;    // [S][stride]  ceil(cdf[s][a] * 2^53), a < A-1 (the last threshold is never reached)
    uint4 *frec = nullptr;      // [S*A][frq]   {Rec of (s,a); top 32 bits of the thr row of the state it leads to}
    uint4 *frec_roll = nullptr; // the same records by ROLLOUT slot (mp_policy_load_ordered), nullptr when the orders agree
    uint32_t *lmask = nullptr;  // policies of STOCHASTIC models (uct_stoch.hip): actions the prior policy lists per state, [S]
    uint8_t *rslot = nullptr;   // ... and the column of every rollout slot, [S][A], nullptr when slots are the columns
    uint8_t *listed8 = nullptr; // ... the listed actions as a byte per action, [S][A], when |A| > 32 (lmask is 32 bits wide)
    // built on the device (policy_build_device): the arrays above are ONE block from the ctx's block cache (hipMalloc / hipFree
    // synchronise the device; a per-episode evaluation loop re-fuses its policy at every step)
    void *block = nullptr;
    size_t block_bytes = 0;
};

namespace mp {

// grow-only device workspace
int ws_reserve(mp_ctx *ctx, int slot, size_t bytes, void **out);

// upload `tab` to WS_TAB0 unless the same table (same kind, same bytes) is already there
int upload_tables(mp_ctx *ctx, int kind, const std::vector<double> &tab, double **dev);

// timing brackets around kernel launches (HIP events on the ctx stream)
int kernels_begin(mp_ctx *ctx);
int kernels_end(mp_ctx *ctx, int n_launches);

template <typename T>
inline int ws_get(mp_ctx *ctx, int slot, size_t count, T **out)
{
    void *p = nullptr;
    int rc = ws_reserve(ctx, slot, count * sizeof(T), &p);
    *out = static_cast<T *>(p);
    return rc;
}

// `mem` of an entry point: bit 0 = the arrays are device arrays; MP_MEM_RNG_DEVICE = rng_state is one even if not
inline int mem_arrays(int mem) { return mem & 1; }
inline int mem_rng(int mem) { return (mem & (MP_MEM_DEVICE | MP_MEM_RNG_DEVICE)) ? MP_MEM_DEVICE : MP_MEM_HOST; }
inline bool mem_valid(int mem) { return mem >= 0 && mem <= 3; }

// uct_stoch.hip: apply a pending re-rooting of the open-loop stochastic trees (mp_uct_step_tree arms it)
} // namespace mp
int uct_stoch_reroot_now(mp_ctx *ctx, long cap_new);
namespace mp {

// (model_index, local root state) pairs of a batch model -> global root states for the planners: `host_tmp` backs *out for
// host arrays (validated), WS_GROOT for device arrays (one small launch on the ctx stream)
int globalize_roots_arg(mp_ctx *ctx, const mp_model *model, int n_roots, const int32_t *model_index, const int32_t *root_state,
                        int mem, std::vector<int32_t> &host_tmp, const int32_t **out);

// side streams for the pipelined host-mode plan: `n` streams forked off the ctx stream / joined back into it
int pipe_fork(mp_ctx *ctx, int n);
int pipe_join(mp_ctx *ctx, int n);

// device alias of a host range inside an mp_host_alloc block (nullptr: not such memory) -- a range check, no HIP call
inline void *pinned_alias(const mp_ctx *ctx, const void *p, size_t bytes)
{
    const char *c = static_cast<const char *>(p);
    for (const auto &b : ctx->pinned)
        if (c >= b.host && c + bytes <= b.host + b.bytes) return b.dev + (c - b.host);
    return nullptr;
}

// copy helper: host->device (sync on the ctx stream) or pass-through of a device pointer; host arrays that live in
// mp_host_alloc memory are passed through as well (the kernels read them over the bus: zero-copy)
template <typename T>
inline int stage_in(mp_ctx *ctx, int slot, const T *src, size_t count, int mem, T **dev)
{
    if (mem == MP_MEM_DEVICE) {
        *dev = const_cast<T *>(src);
        return MP_OK;
    }
    if (src)
        if (void *alias = pinned_alias(ctx, src, count * sizeof(T))) {
            *dev = static_cast<T *>(alias);
            return MP_OK;
        }
    MP_TRY(ws_get(ctx, slot, count, dev));
    if (src) MP_HIP(hipMemcpyAsync(*dev, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return MP_OK;
}

template <typename T>
inline int stage_out_alloc(mp_ctx *ctx, int slot, T *dst, size_t count, int mem, T **dev)
{
    if (mem == MP_MEM_DEVICE) {
        *dev = dst; // may be nullptr: kernels skip null outputs
        return MP_OK;
    }
    if (!dst) {
        *dev = nullptr;
        return MP_OK;
    }
    if (void *alias = pinned_alias(ctx, dst, count * sizeof(T))) { // zero-copy: the kernel writes the caller's array
        *dev = static_cast<T *>(alias);
        return MP_OK;
    }
    return ws_get(ctx, slot, count, dev);
}

template <typename T>
inline int stage_out_copy(mp_ctx *ctx, T *dst, const T *dev, size_t count, int mem)
{
    if (mem == MP_MEM_DEVICE || !dst || !dev) return MP_OK;
    if (pinned_alias(ctx, dst, count * sizeof(T)) == static_cast<const void *>(dev)) return MP_OK; // written in place
    MP_HIP(hipMemcpyAsync(dst, dev, count * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    return MP_OK;
}

} // namespace mp
