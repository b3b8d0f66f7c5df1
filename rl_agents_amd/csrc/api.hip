// api.hip -- context, transition-model upload and host helpers of libmi355plan.so.
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <set>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "libm_sincos.hpp"

namespace mp {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int ws_reserve(mp_ctx *ctx, int slot, size_t bytes, void **out)
{
    DevBuf &b = ctx->ws[slot];
    if (bytes > b.cap) {
        if (b.p) {
            // buffers may still be referenced by enqueued work
            MP_HIP(hipStreamSynchronize(ctx->stream));
            MP_HIP(hipFree(b.p));
            b.p = nullptr;
            b.cap = 0;
        }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&b.p, want);
        if (e != hipSuccess && !ctx->block_cache.empty()) { // dead planner blocks may be what is in the way
            (void)hipGetLastError();
            ctx_block_cache_flush(ctx);
            e = hipMalloc(&b.p, want);
        }
        if (e != hipSuccess) {
            b.p = nullptr;
            return fail(MP_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        }
        b.cap = want;
    }
    *out = b.p;
    return MP_OK;
}

int upload_tables(mp_ctx *ctx, int kind, const std::vector<double> &tab, double **dev)
{
    double *d = nullptr;
    const bool same = ctx->tab_kind == kind && ctx->tab_host.size() == tab.size() && ctx->ws[WS_TAB0].p &&
                      memcmp(ctx->tab_host.data(), tab.data(), tab.size() * sizeof(double)) == 0;
    MP_TRY(ws_get(ctx, WS_TAB0, tab.size(), &d));
    if (!same) {
        ctx->tab_kind = 0;
        MP_HIP(hipMemcpyAsync(d, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        MP_HIP(hipStreamSynchronize(ctx->stream)); // pageable source: must be consumed before return
        ctx->tab_host = tab;
        ctx->tab_kind = kind;
    }
    *dev = d;
    return MP_OK;
}

int kernels_begin(mp_ctx *ctx)
{
    MP_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return MP_OK;
}

int kernels_end(mp_ctx *ctx, int n_launches)
{
    MP_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timed = true;
    ctx->n_launches = n_launches;
    return MP_OK;
}

int pipe_fork(mp_ctx *ctx, int n)
{
    if (n < 1 || n > 8) return fail(MP_ERR_ARG, "pipe_fork: %d streams", n);
    if (!ctx->pipe_fork) MP_HIP(hipEventCreateWithFlags(&ctx->pipe_fork, hipEventDisableTiming));
    MP_HIP(hipEventRecord(ctx->pipe_fork, ctx->stream));
    for (int i = 0; i < n; ++i) {
        if (!ctx->pipe[i]) {
            MP_HIP(hipStreamCreateWithFlags(&ctx->pipe[i], hipStreamNonBlocking));
            MP_HIP(hipEventCreateWithFlags(&ctx->pipe_done[i], hipEventDisableTiming));
        }
        MP_HIP(hipStreamWaitEvent(ctx->pipe[i], ctx->pipe_fork, 0));
    }
    return MP_OK;
}

int pipe_join(mp_ctx *ctx, int n)
{
    for (int i = 0; i < n; ++i) {
        MP_HIP(hipEventRecord(ctx->pipe_done[i], ctx->pipe[i]));
        MP_HIP(hipStreamWaitEvent(ctx->stream, ctx->pipe_done[i], 0));
    }
    return MP_OK;
}

// ------------------------------------------------------------------ numpy SeedSequence -> PCG64 ---
// numpy/random/bit_generator.pyx SeedSequence (pool of 4 uint32 words) and PCG64's seeding from
// generate_state(4, uint64), restated; tests/test_host_logic.py compares the records with numpy's.
namespace {
constexpr uint32_t kInitA = 0x43b0d7e5u, kMultA = 0x931e8875u, kInitB = 0x8b51f9ddu, kMultB = 0x58f38dedu;
constexpr uint32_t kMixL = 0xca01f9ddu, kMixR = 0x4973f715u;
inline uint32_t ss_hashmix(uint32_t value, uint32_t &hash_const)
{
    value ^= hash_const;
    hash_const *= kMultA;
    value *= hash_const;
    value ^= value >> 16;
    return value;
}
inline uint32_t ss_mix(uint32_t x, uint32_t y)
{
    uint32_t r = kMixL * x - kMixR * y;
    r ^= r >> 16;
    return r;
}
// entropy words (already split into uint32, no spawn key) -> the six-word PCG64 record
void seed_sequence_record(const uint32_t *entropy, int n, uint64_t *rec)
{
    uint32_t pool[4], hc = kInitA;
    for (int i = 0; i < 4; ++i) pool[i] = ss_hashmix(i < n ? entropy[i] : 0u, hc);
    for (int src = 0; src < 4; ++src)
        for (int dst = 0; dst < 4; ++dst)
            if (src != dst) pool[dst] = ss_mix(pool[dst], ss_hashmix(pool[src], hc));
    for (int src = 4; src < n; ++src)
        for (int dst = 0; dst < 4; ++dst) pool[dst] = ss_mix(pool[dst], ss_hashmix(entropy[src], hc));
    uint32_t w[8], hb = kInitB;
    for (int i = 0; i < 8; ++i) {          // generate_state(4, uint64) = 8 uint32 words viewed as 4 little-endian uint64
        uint32_t v = pool[i & 3];
        v ^= hb;
        hb *= kMultB;
        v *= hb;
        v ^= v >> 16;
        w[i] = v;
    }
    uint64_t q[4];
    for (int i = 0; i < 4; ++i) q[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    typedef unsigned __int128 u128;
    const u128 mult = ((u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
    const u128 initstate = ((u128)q[0] << 64) | q[1], initseq = ((u128)q[2] << 64) | q[3];
    const u128 inc = (initseq << 1) | 1u;        // pcg_setseq_128_srandom_r
    u128 state = 0;
    state = state * mult + inc;
    state += initstate;
    state = state * mult + inc;
    rec[0] = (uint64_t)(state >> 64); rec[1] = (uint64_t)state;
    rec[2] = (uint64_t)(inc >> 64); rec[3] = (uint64_t)inc;
    rec[4] = 0; rec[5] = 0;                      // has_uint32, uinteger
}
// numpy's _int_to_uint32_array: little-endian 32-bit words of a non-negative integer, [0] for 0
inline int key_words(int64_t key, uint32_t *out)
{
    uint64_t k = (uint64_t)key;
    out[0] = (uint32_t)k;
    if (!(k >> 32)) return 1;
    out[1] = (uint32_t)(k >> 32);
    return 2;
}
} // namespace

// packs T/R/terminal of model 0 into 16-byte records
__global__ void pack_records(int S, int A, const int32_t *__restrict__ T, const double *__restrict__ R,
                             const uint8_t *__restrict__ term, const uint8_t *__restrict__ avail, Rec *__restrict__ rec)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)S * A) return;
    const int s = (int)(i / A);
    Rec r;
    r.next = T[i];
    r.flags = (term && term[s] ? 1u : 0u) | (term && term[r.next] ? 2u : 0u) | (!avail || avail[i] ? 4u : 0u);
    r.reward = R[i];
    rec[i] = r;
}

// the same for the records of states [s0, s0 + ns) only (mp_model_update_tables / _rows: T, term, avail, rec are the
// model's whole arrays)
__global__ void pack_records_range(int s0, int ns, int A, const int32_t *__restrict__ T, const double *__restrict__ R,
                                   const uint8_t *__restrict__ term, const uint8_t *__restrict__ avail, Rec *__restrict__ rec,
                                   uint16_t *__restrict__ t16)
{
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (long)ns * A) return;
    const long i = (long)s0 * A + j;
    const int s = (int)(i / A);
    Rec r;
    r.next = T[i];
    r.flags = (term && term[s] ? 1u : 0u) | (term && term[r.next] ? 2u : 0u) | (!avail || avail[i] ? 4u : 0u);
    r.reward = R[i];
    rec[i] = r;
    if (t16) t16[i] = (uint16_t)((uint32_t)r.next | (term && term[r.next] ? 0x8000u : 0u));
}

// mp_model_update_rows: rows[k] = global state id; its |A| transitions / rewards (+ terminal flag, + reward indices) from the
// staged arrays into the model's tables
__global__ void scatter_rows(int n_rows, int A, const int32_t *__restrict__ rows, const int32_t *__restrict__ t_new,
                             const double *__restrict__ r_new, const uint8_t *__restrict__ term_new,
                             const uint8_t *__restrict__ r8_new, int32_t *__restrict__ T, double *__restrict__ R,
                             uint8_t *__restrict__ term, uint8_t *__restrict__ r8)
{
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (long)n_rows * A) return;
    const int k = (int)(j / A), a = (int)(j - (long)k * A);
    const long i = (long)rows[k] * A + a;
    T[i] = t_new[j];
    R[i] = r_new[j];
    if (r8 && r8_new) r8[i] = r8_new[j];
    if (a == 0 && term && term_new) term[rows[k]] = term_new[k];
}

// ... and the records of exactly those rows (terminal flags unchanged)
__global__ void pack_records_rows(int n_rows, int A, const int32_t *__restrict__ rows, const int32_t *__restrict__ T,
                                  const double *__restrict__ R, const uint8_t *__restrict__ term, const uint8_t *__restrict__ avail,
                                  Rec *__restrict__ rec, uint16_t *__restrict__ t16)
{
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (long)n_rows * A) return;
    const int k = (int)(j / A), a = (int)(j - (long)k * A);
    const int s = rows[k];
    const long i = (long)s * A + a;
    Rec r;
    r.next = T[i];
    r.flags = (term && term[s] ? 1u : 0u) | (term && term[r.next] ? 2u : 0u) | (!avail || avail[i] ? 4u : 0u);
    r.reward = R[i];
    rec[i] = r;
    if (t16) t16[i] = (uint16_t)((uint32_t)r.next | (term && term[r.next] ? 0x8000u : 0u));
}

// root i of a batch model: global state = model_index[i] * Sb + local state (mp_uct_plan_models / mp_opd_plan_models)
// Device arrays cannot be validated on the host: a root naming a model outside [0, NB) or a state outside [0, Sb) is CLAMPED to
// state 0 of model 0 (the planners then read nothing out of bounds) and counted in the ctx's sticky fault word.
__global__ void globalize_roots(int n, int Sb, int NB, const int32_t *__restrict__ model_index, const int32_t *__restrict__ local,
                                int32_t *__restrict__ global, int32_t *__restrict__ fault)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int m = model_index[i], s = local[i];
    const bool bad = m < 0 || m >= NB || s < 0 || s >= Sb;
    if (bad && fault) atomicAdd(fault, 1);
    global[i] = bad ? 0 : m * Sb + s;
}

// compact transitions for the LDS variant of the UCT kernel (S < 32768)
__global__ void pack_t16(int S, int A, const int32_t *__restrict__ T, const uint8_t *__restrict__ term,
                         uint16_t *__restrict__ t16)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)S * A) return;
    const int nx = T[i];
    t16[i] = (uint16_t)((uint32_t)nx | (term && term[nx] ? 0x8000u : 0u));
}

// trainer/evaluation.py:164-190 for a batch of episodes of one table model (see mi355plan.h)
__global__ void env_step_kernel(int n, int A, const Rec *__restrict__ rec, int done_on_next, int32_t *__restrict__ state,
                                int32_t *__restrict__ steps, uint8_t *__restrict__ alive, const int32_t *__restrict__ plans,
                                int plan_stride, int max_steps, const double *__restrict__ gpow, double *__restrict__ returns,
                                double *__restrict__ discounted, int32_t *__restrict__ actions_log, int log_stride,
                                int32_t *__restrict__ n_alive)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = false;
    if (i < n && alive[i]) {
        const int planned = plans[(long)i * plan_stride];
        const int act = planned < 0 ? 0 : planned; // an empty plan: logged as -1 (the caller raises, as Evaluation.step does)
        const int s = state[i], t = steps[i];
        const Rec rc = rec[(long)s * A + act];
        const bool done = (rc.flags & (done_on_next ? 2u : 1u)) != 0;
        returns[i] += rc.reward;
        discounted[i] += rc.reward * gpow[t];
        if (actions_log && t < log_stride) actions_log[(long)i * log_stride + t] = planned < 0 ? -1 : act;
        state[i] = rc.next;
        steps[i] = t + 1;
        live = !(done || t + 1 >= max_steps);
        alive[i] = live ? 1 : 0;
    }
    const unsigned long long b = __ballot(live);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_alive, (int)__popcll(b));
}

__global__ void greedy_actions_kernel(int n, int A, const double *__restrict__ Q, const int32_t *__restrict__ state,
                                      int32_t *__restrict__ plans, int plan_stride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *q = Q + (long)state[i] * A;
    int best = 0;
    double m = q[0];
    for (int a = 1; a < A; ++a)
        if (q[a] > m) { m = q[a]; best = a; } // np.argmax: first maximum
    plans[(long)i * plan_stride] = best;
}

// mp_pack_rows / mp_unpack_rows (mi355plan.h): the per-root results of a sharded plan <-> ONE byte matrix for the collective.
// One thread per 4-byte word of the packed matrix: the packed side is a coalesced stream, the array side runs of
// width / 4 words.
struct RowCols {
    const void *ptr[8];
    int off[9]; // word offset of array k inside a packed row; off[n] = words per row
    int n;
};

__global__ void pack_rows_kernel(RowCols c, int n_local, int per, uint32_t *__restrict__ packed)
{
    const int wpr = c.off[c.n];
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)per * wpr) return;
    const int row = (int)(t / wpr), col = (int)(t - (long)row * wpr);
    uint32_t v = 0;
    if (row < n_local) {
        int k = 0;
        while (k + 1 < c.n && col >= c.off[k + 1]) ++k;
        const int w = c.off[k + 1] - c.off[k];
        v = static_cast<const uint32_t *>(c.ptr[k])[(long)row * w + (col - c.off[k])];
    }
    packed[t] = v;
}

__global__ void unpack_rows_kernel(RowCols c, int n_total, int world, int per, const uint32_t *__restrict__ packed)
{
    const int wpr = c.off[c.n];
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)n_total * wpr) return;
    const int row = (int)(t / wpr), col = (int)(t - (long)row * wpr);
    // rank and local row of global row `row` under the balanced contiguous split (distributed.shard_bounds)
    const int base = n_total / world, extra = n_total - base * world;
    int r, j;
    if (row < extra * (base + 1)) { r = row / (base + 1); j = row - r * (base + 1); }
    else { r = extra + (row - extra * (base + 1)) / base; j = (row - extra * (base + 1)) - (r - extra) * base; }
    int k = 0;
    while (k + 1 < c.n && col >= c.off[k + 1]) ++k;
    const int w = c.off[k + 1] - c.off[k];
    static_cast<uint32_t *>(const_cast<void *>(c.ptr[k]))[(long)row * w + (col - c.off[k])] = packed[((long)r * per + j) * wpr + col];
}

int globalize_roots_arg(mp_ctx *ctx, const mp_model *model, int n_roots, const int32_t *model_index, const int32_t *root_state,
                        int mem, std::vector<int32_t> &host_tmp, const int32_t **out)
{
    if (!ctx || !model || !model_index || !root_state) return fail(MP_ERR_ARG, "plan on a batch model: NULL argument");
    if (model->mode != MP_MODE_DETERMINISTIC || !model->rec) return fail(MP_ERR_MODE, "plan on a batch model: deterministic table models only");
    if (n_roots < 1) return fail(MP_ERR_ARG, "plan on a batch model: n_roots = %d", n_roots);
    const int Sb = model->Sb > 0 ? model->Sb : model->S;
    if (mem_arrays(mem) == MP_MEM_DEVICE) {
        int32_t *d = nullptr;
        MP_HIP(hipSetDevice(ctx->device));
        MP_TRY(ws_get(ctx, WS_GROOT, (size_t)n_roots, &d));
        if (!ctx->fault_host) {
            MP_HIP(hipHostMalloc((void **)&ctx->fault_host, 64, hipHostMallocMapped));
            *ctx->fault_host = 0;
            MP_HIP(hipHostGetDevicePointer((void **)&ctx->fault_dev, ctx->fault_host, 0));
        }
        hipLaunchKernelGGL(globalize_roots, dim3((unsigned)((n_roots + 255) / 256)), dim3(256), 0, ctx->stream, n_roots, Sb,
                           model->NB > 0 ? model->NB : 1, model_index, root_state, d, ctx->fault_dev);
        MP_HIP(hipGetLastError());
        *out = d;
        return MP_OK;
    }
    host_tmp.resize((size_t)n_roots);
    for (int i = 0; i < n_roots; ++i) {
        if (model_index[i] < 0 || model_index[i] >= model->NB)
            return fail(MP_ERR_ARG, "plan on a batch model: model_index[%d] = %d outside [0, %d)", i, model_index[i], model->NB);
        if (root_state[i] < 0 || root_state[i] >= Sb)
            return fail(MP_ERR_ARG, "plan on a batch model: root_state[%d] = %d outside [0, %d)", i, root_state[i], Sb);
        host_tmp[i] = model_index[i] * Sb + root_state[i];
    }
    *out = host_tmp.data();
    return MP_OK;
}

} // namespace mp

using namespace mp;

extern "C" {

const char *mp_last_error(void) { return g_err.c_str(); }
int mp_abi_version(void) { return MP_ABI_VERSION; }

extern "C++" {
namespace {
std::mutex g_ctx_mutex;
std::set<const mp_ctx *> g_live_ctx;
} // namespace

bool mp_ctx_alive(const mp_ctx *ctx)
{
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    return g_live_ctx.count(ctx) != 0;
}
} // extern "C++"

int mp_ctx_create(int device, void *stream, mp_ctx **out)
{
    if (!out) return fail(MP_ERR_ARG, "mp_ctx_create: out is NULL");
    int n = 0;
    MP_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(MP_ERR_ARG, "mp_ctx_create: device %d out of range (%d visible)", device, n);
    MP_HIP(hipSetDevice(device));
    mp_ctx *ctx = new mp_ctx();
    ctx->device = device;
    if (hipGetDeviceProperties(&ctx->prop, device) != hipSuccess) {
        delete ctx;
        return fail(MP_ERR_HIP, "hipGetDeviceProperties failed");
    }
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return fail(MP_ERR_HIP, "hipStreamCreate failed");
        }
        ctx->own_stream = true;
    }
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return fail(MP_ERR_HIP, "hipEventCreate failed");
    }
    { std::lock_guard<std::mutex> lock(g_ctx_mutex); g_live_ctx.insert(ctx); }
    *out = ctx;
    return MP_OK;
}

int mp_env_step(mp_ctx *ctx, mp_model *model, int32_t n, int32_t *state, int32_t *steps, uint8_t *alive,
                const int32_t *plans, int32_t plan_stride, int32_t max_steps, const double *gpow, double *returns,
                double *discounted, int32_t *actions_log, int32_t log_stride, int32_t *n_alive, int32_t mem)
{
    if (!ctx || !model || !state || !steps || !alive || !plans || !gpow || !returns || !discounted || !n_alive)
        return fail(MP_ERR_ARG, "mp_env_step: NULL argument");
    if (mem != MP_MEM_DEVICE) return fail(MP_ERR_ARG, "mp_env_step: device arrays only (the loop it serves has no host side)");
    if (model->mode != MP_MODE_DETERMINISTIC || !model->rec) return fail(MP_ERR_MODE, "mp_env_step: deterministic table models only");
    if (n < 1 || plan_stride < 1 || max_steps < 1) return fail(MP_ERR_ARG, "mp_env_step: bad sizes");
    MP_HIP(hipSetDevice(ctx->device));
    MP_HIP(hipMemsetAsync(n_alive, 0, sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(env_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, model->A, model->rec,
                       model->done_on_next, state, steps, alive, plans, plan_stride, max_steps, gpow, returns, discounted,
                       actions_log, log_stride, n_alive);
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_greedy_actions(mp_ctx *ctx, int32_t n, int32_t S, int32_t A, const double *Q, const int32_t *state, int32_t *plans,
                      int32_t plan_stride, int32_t mem)
{
    if (!ctx || !Q || !state || !plans) return fail(MP_ERR_ARG, "mp_greedy_actions: NULL argument");
    if (mem != MP_MEM_DEVICE) return fail(MP_ERR_ARG, "mp_greedy_actions: device arrays only");
    if (n < 1 || S < 1 || A < 1 || plan_stride < 1) return fail(MP_ERR_ARG, "mp_greedy_actions: bad sizes");
    MP_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(greedy_actions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, A, Q, state, plans,
                       plan_stride);
    MP_HIP(hipGetLastError());
    return MP_OK;
}

static int row_cols(const char *who, int32_t n_arrays, const void *const *ptr, const int32_t *width, RowCols *c)
{
    if (n_arrays < 1 || n_arrays > 8 || !ptr || !width) return fail(MP_ERR_ARG, "%s: 1..8 arrays", who);
    c->n = n_arrays;
    c->off[0] = 0;
    for (int k = 0; k < n_arrays; ++k) {
        if (!ptr[k] || width[k] < 4 || (width[k] & 3)) return fail(MP_ERR_ARG, "%s: array %d: NULL or row width %d not a positive multiple of 4", who, k, width[k]);
        c->ptr[k] = ptr[k];
        c->off[k + 1] = c->off[k] + width[k] / 4;
    }
    for (int k = n_arrays; k < 8; ++k) c->ptr[k] = nullptr;
    return MP_OK;
}

int mp_pack_rows(mp_ctx *ctx, void *stream, int32_t n_local, int32_t per, int32_t n_arrays, const void *const *src,
                 const int32_t *width, void *packed)
{
    if (!ctx || !packed) return fail(MP_ERR_ARG, "mp_pack_rows: NULL argument");
    if (n_local < 0 || per < 1 || n_local > per) return fail(MP_ERR_ARG, "mp_pack_rows: bad sizes (n_local=%d per=%d)", n_local, per);
    RowCols c;
    MP_TRY(row_cols("mp_pack_rows", n_arrays, src, width, &c));
    MP_HIP(hipSetDevice(ctx->device));
    const long words = (long)per * c.off[c.n];
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream ? (hipStream_t)stream : ctx->stream,
                       c, n_local, per, static_cast<uint32_t *>(packed));
    MP_HIP(hipGetLastError());
    return MP_OK;
}

int mp_unpack_rows(mp_ctx *ctx, void *stream, int32_t n_total, int32_t world, int32_t per, int32_t n_arrays,
                   const void *packed, const int32_t *width, void *const *dst)
{
    if (!ctx || !packed) return fail(MP_ERR_ARG, "mp_unpack_rows: NULL argument");
    if (world < 1 || n_total < world || (long)per * world < n_total || per < (n_total + world - 1) / world)
        return fail(MP_ERR_ARG, "mp_unpack_rows: bad sizes (n_total=%d world=%d per=%d)", n_total, world, per);
    RowCols c;
    MP_TRY(row_cols("mp_unpack_rows", n_arrays, dst, width, &c));
    MP_HIP(hipSetDevice(ctx->device));
    const long words = (long)n_total * c.off[c.n];
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream ? (hipStream_t)stream : ctx->stream,
                       c, n_total, world, per, static_cast<const uint32_t *>(packed));
    MP_HIP(hipGetLastError());
    return MP_OK;
}

// ---- RCCL, resolved at run time (mi355plan.h: mp_comm_*) ------------------------------------------------------------------
extern "C++" {
namespace {
struct RcclUid { char b[MP_COMM_UID_BYTES]; };   // ncclUniqueId: 128 bytes, passed BY VALUE to ncclCommInitRank
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, RcclUid, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int rccl_load()
{
    if (g_rccl.lib) return MP_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);          // the copy the process already holds (PyTorch's), if any
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(MP_ERR_HIP, "mp_comm: librccl.so.1 not found (%s)", dlerror());
    Rccl r;
    r.lib = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy)
        return fail(MP_ERR_HIP, "mp_comm: librccl.so.1 lacks an expected symbol");
    g_rccl = r;
    return MP_OK;
}
int rccl_fail(const char *what, int rc)
{
    return fail(MP_ERR_HIP, "%s failed: %s (ncclResult %d)", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?", rc);
}
} // namespace
} // extern "C++"

int mp_comm_unique_id(void *uid)
{
    if (!uid) return fail(MP_ERR_ARG, "mp_comm_unique_id: uid is NULL");
    MP_TRY(rccl_load());
    const int rc = g_rccl.GetUniqueId(uid);
    return rc == 0 ? MP_OK : rccl_fail("ncclGetUniqueId", rc);
}

int mp_comm_destroy(mp_ctx *ctx)
{
    if (!ctx) return fail(MP_ERR_ARG, "mp_comm_destroy: ctx is NULL");
    if (ctx->comm && g_rccl.CommDestroy) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)g_rccl.CommDestroy(ctx->comm);
    }
    ctx->comm = nullptr;
    ctx->comm_world = 0;
    return MP_OK;
}

int mp_comm_init(mp_ctx *ctx, int32_t rank, int32_t world, const void *uid)
{
    if (!ctx || !uid) return fail(MP_ERR_ARG, "mp_comm_init: NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(MP_ERR_ARG, "mp_comm_init: rank %d of %d", rank, world);
    MP_TRY(rccl_load());
    MP_HIP(hipSetDevice(ctx->device));
    if (ctx->comm) MP_TRY(mp_comm_destroy(ctx));
    RcclUid id;
    memcpy(id.b, uid, MP_COMM_UID_BYTES);
    void *comm = nullptr;
    const int rc = g_rccl.CommInitRank(&comm, world, id, rank);
    if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
    ctx->comm = comm;
    ctx->comm_world = world;
    return MP_OK;
}

int mp_gather_results(mp_ctx *ctx, void *stream, int32_t per, int32_t row_bytes, const void *packed, void *gathered)
{
    if (!ctx || !packed || !gathered) return fail(MP_ERR_ARG, "mp_gather_results: NULL argument");
    if (!ctx->comm) return fail(MP_ERR_ARG, "mp_gather_results: no communicator on this ctx (mp_comm_init)");
    if (per < 1 || row_bytes < 1) return fail(MP_ERR_ARG, "mp_gather_results: per %d, row_bytes %d", per, row_bytes);
    MP_HIP(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int rc = g_rccl.AllGather(packed, gathered, (size_t)per * row_bytes, /* ncclInt8 */ 0, ctx->comm, s);
    return rc == 0 ? MP_OK : rccl_fail("ncclAllGather", rc);
}

int mp_seed_sequence_states(const uint32_t *entropy, int32_t n_words, int64_t first_key, int32_t count, uint64_t *out)
{
    if (!out || n_words < 0 || (n_words && !entropy) || count < 0 || first_key < 0)
        return fail(MP_ERR_ARG, "mp_seed_sequence_states: bad argument");
    std::vector<uint32_t> e((size_t)n_words + 2);
    for (int i = 0; i < n_words; ++i) e[i] = entropy[i];
    for (int i = 0; i < count; ++i) {
        const int k = key_words(first_key + i, e.data() + n_words);
        seed_sequence_record(e.data(), n_words + k, out + (size_t)i * 6);
    }
    return MP_OK;
}

int mp_host_alloc(mp_ctx *ctx, int64_t bytes, void **out)
{
    if (!ctx || !out || bytes < 0) return fail(MP_ERR_ARG, "mp_host_alloc: bad argument");
    MP_HIP(hipSetDevice(ctx->device));
    void *p = nullptr;
    const size_t n = (size_t)(bytes > 0 ? bytes : 1);
    if (hipHostMalloc(&p, n, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess)
        return fail(MP_ERR_ALLOC, "mp_host_alloc: hipHostMalloc(%lld) failed", (long long)bytes);
    void *d = nullptr;
    if (!getenv("MP_NO_ZERO_COPY") && hipHostGetDevicePointer(&d, p, 0) == hipSuccess && d)
        ctx->pinned.push_back({static_cast<const char *>(p), static_cast<char *>(d), n});
    *out = p;
    return MP_OK;
}

int mp_host_free(mp_ctx *ctx, void *ptr)
{
    if (!ctx) return fail(MP_ERR_ARG, "mp_host_free: ctx is NULL");
    if (!ptr) return MP_OK;
    MP_HIP(hipSetDevice(ctx->device));
    MP_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < ctx->pinned.size(); ++i)
        if (ctx->pinned[i].host == ptr) { ctx->pinned.erase(ctx->pinned.begin() + (long)i); break; }
    MP_HIP(hipHostFree(ptr));
    return MP_OK;
}

int mp_rng_create(mp_ctx *ctx, int32_t n, mp_rng **out)
{
    if (!ctx || !out || n < 1) return fail(MP_ERR_ARG, "mp_rng_create: bad argument");
    MP_HIP(hipSetDevice(ctx->device));
    mp_rng *r = new (std::nothrow) mp_rng;
    if (!r) return fail(MP_ERR_ALLOC, "mp_rng_create: out of memory");
    r->ctx = ctx; r->n = n;
    if (hipMalloc(&r->state, (size_t)n * 48) != hipSuccess) {
        delete r;
        return fail(MP_ERR_ALLOC, "mp_rng_create: hipMalloc(%zu) failed", (size_t)n * 48);
    }
    *out = r;
    return MP_OK;
}

int mp_rng_free(mp_rng *rng)
{
    if (!rng) return MP_OK;
    hipSetDevice(rng->ctx->device);
    hipStreamSynchronize(rng->ctx->stream);
    if (rng->state) (void)hipFree(rng->state);
    delete rng;
    return MP_OK;
}

int mp_rng_set(mp_rng *rng, int32_t first, int32_t count, const uint64_t *state6)
{
    if (!rng || !state6 || first < 0 || count < 0 || (long)first + count > rng->n) return fail(MP_ERR_ARG, "mp_rng_set: bad range");
    MP_HIP(hipSetDevice(rng->ctx->device));
    MP_HIP(hipMemcpyAsync(rng->state + (size_t)first * 6, state6, (size_t)count * 48, hipMemcpyHostToDevice, rng->ctx->stream));
    MP_HIP(hipStreamSynchronize(rng->ctx->stream));
    return MP_OK;
}

int mp_rng_get(mp_rng *rng, int32_t first, int32_t count, uint64_t *state6)
{
    if (!rng || !state6 || first < 0 || count < 0 || (long)first + count > rng->n) return fail(MP_ERR_ARG, "mp_rng_get: bad range");
    MP_HIP(hipSetDevice(rng->ctx->device));
    MP_HIP(hipMemcpyAsync(state6, rng->state + (size_t)first * 6, (size_t)count * 48, hipMemcpyDeviceToHost, rng->ctx->stream));
    MP_HIP(hipStreamSynchronize(rng->ctx->stream));
    return MP_OK;
}

int mp_rng_seed_sequence(mp_rng *rng, int32_t first, int32_t count, const uint32_t *entropy, int32_t n_words, int64_t first_key)
{
    if (!rng || first < 0 || count < 0 || (long)first + count > rng->n) return fail(MP_ERR_ARG, "mp_rng_seed_sequence: bad range");
    std::vector<uint64_t> h((size_t)count * 6);
    MP_TRY(mp_seed_sequence_states(entropy, n_words, first_key, count, h.data()));
    return mp_rng_set(rng, first, count, h.data());
}

uint64_t *mp_rng_device_ptr(mp_rng *rng, int32_t first)
{
    if (!rng || first < 0 || first >= rng->n) {
        fail(MP_ERR_ARG, "mp_rng_device_ptr: bad argument");
        return nullptr;
    }
    return rng->state + (size_t)first * 6;
}

int mp_ctx_destroy(mp_ctx *ctx)
{
    if (ctx && ctx->comm) (void)mp_comm_destroy(ctx);
    if (!ctx) return MP_OK;
    { std::lock_guard<std::mutex> lock(g_ctx_mutex); g_live_ctx.erase(ctx); }
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 8; ++i) {
        if (ctx->pipe[i]) { hipStreamSynchronize(ctx->pipe[i]); hipStreamDestroy(ctx->pipe[i]); }
        if (ctx->pipe_done[i]) hipEventDestroy(ctx->pipe_done[i]);
    }
    if (ctx->pipe_fork) hipEventDestroy(ctx->pipe_fork);
    if (ctx->vi_graph_exec) hipGraphExecDestroy((hipGraphExec_t)ctx->vi_graph_exec);
    for (auto &b : ctx->ws)
        if (b.p) hipFree(b.p);
    for (auto &b : ctx->block_cache) hipFree(b.p);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->fault_host) hipHostFree(ctx->fault_host);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return MP_OK;
}

int mp_ctx_set_stream(mp_ctx *ctx, void *stream)
{
    if (!ctx) return fail(MP_ERR_ARG, "ctx is NULL");
    MP_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) {
        MP_HIP(hipStreamDestroy(ctx->stream));
        ctx->own_stream = false;
    }
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        MP_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return MP_OK;
}

int mp_ctx_get_stream(mp_ctx *ctx, void **stream)
{
    if (!ctx || !stream) return fail(MP_ERR_ARG, "mp_ctx_get_stream: NULL argument");
    *stream = (void *)ctx->stream;
    return MP_OK;
}

int mp_ctx_synchronize(mp_ctx *ctx)
{
    if (!ctx) return fail(MP_ERR_ARG, "ctx is NULL");
    MP_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->fault_host && *ctx->fault_host) {
        const int n = *ctx->fault_host;
        *ctx->fault_host = 0;
        return fail(MP_ERR_ARG, "%d root(s) of a device-array plan on a batch model named a model or state out of range "
                                "(planned from state 0 of model 0 instead): their results are meaningless", n);
    }
    return MP_OK;
}

int mp_ctx_device_faults(mp_ctx *ctx, int32_t *count)
{
    if (!ctx || !count) return fail(MP_ERR_ARG, "mp_ctx_device_faults: NULL argument");
    *count = ctx->fault_host ? *ctx->fault_host : 0;
    return MP_OK;
}

int mp_ctx_device_info(mp_ctx *ctx, int32_t *n_cu, int32_t *wave_size, int64_t *lds_bytes, int64_t *hbm_bytes,
                       char *name, int32_t name_cap)
{
    if (!ctx) return fail(MP_ERR_ARG, "ctx is NULL");
    if (n_cu) *n_cu = ctx->prop.multiProcessorCount;
    if (wave_size) *wave_size = ctx->prop.warpSize;
    if (lds_bytes) *lds_bytes = (int64_t)ctx->prop.maxSharedMemoryPerMultiProcessor;
    if (hbm_bytes) *hbm_bytes = (int64_t)ctx->prop.totalGlobalMem;
    if (name && name_cap > 0) {
        snprintf(name, (size_t)name_cap, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    }
    return MP_OK;
}

const char *mp_last_kernel_variant(mp_ctx *ctx) { return ctx ? ctx->last_variant : ""; }

// ---- self-test: lane order of same-address LDS atomics within one wave instruction (see mi355plan.h) ----------------------
namespace mp {
__device__ __forceinline__ uint32_t st_hash(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(64) void selftest_lds_order_kernel(int waves, unsigned long long *violations)
{
    __shared__ double cell[8];
    const int lane = threadIdx.x;
    unsigned long long bad = 0;
    for (int w = blockIdx.x; w < waves; w += gridDim.x) {
        if (lane < 8) cell[lane] = 5.0;
        __syncthreads();
        const uint32_t h = st_hash((uint32_t)w * 64u + (uint32_t)lane);
        const int n_addr = (w & 7) + 1;
        const int a = (int)((h >> 8) % (uint32_t)n_addr);
        const double v = (double)(h & 63u) / 8.0;
        const bool on = (w & 3) == 0 || ((h >> 20) & 1u);
        double got = -1.0;
        if (on) got = __hip_atomic_fetch_min(&cell[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        // what lane order prescribes: the minimum of 5.0 and the values of the LOWER active lanes on the same cell
        double want = 5.0, fin = 5.0;
        for (int l = 0; l < 64; ++l) {
            const uint32_t hl = st_hash((uint32_t)w * 64u + (uint32_t)l);
            const bool on_l = (w & 3) == 0 || ((hl >> 20) & 1u);
            const int a_l = (int)((hl >> 8) % (uint32_t)n_addr);
            const double v_l = (double)(hl & 63u) / 8.0;
            if (on_l && a_l == a) {
                if (l < lane) want = v_l < want ? v_l : want;
                fin = v_l < fin ? v_l : fin;
            }
        }
        if (on && got != want) ++bad;
        if (on && cell[a] != fin) ++bad;
        __syncthreads();
    }
    if (bad) atomicAdd(violations, bad);
}
} // namespace mp

int mp_selftest_lds_atomic_order(mp_ctx *ctx, int32_t waves, int64_t *violations)
{
    if (!ctx || !violations || waves < 1) return fail(MP_ERR_ARG, "mp_selftest_lds_atomic_order: bad argument");
    MP_HIP(hipSetDevice(ctx->device));
    unsigned long long *d = nullptr;
    MP_HIP(hipMalloc(&d, 8));
    unsigned long long h = 0;
    hipError_t e = hipMemsetAsync(d, 0, 8, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(mp::selftest_lds_order_kernel, dim3((unsigned)(waves < 4096 ? waves : 4096)), dim3(64), 0, ctx->stream, waves, d);
        e = hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, ctx->stream);
    }
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d); // (freed on every path)
    MP_HIP(e);
    MP_HIP(e2);
    *violations = (int64_t)h;
    return MP_OK;
}

int mp_last_kernel_ms(mp_ctx *ctx, double *ms, int32_t *n_launches)
{
    if (!ctx) return fail(MP_ERR_ARG, "ctx is NULL");
    if (!ctx->timed) return fail(MP_ERR_ARG, "no timed kernel batch on this ctx yet");
    MP_HIP(hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    MP_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    if (ms) *ms = (double)f;
    if (n_launches) *n_launches = ctx->n_launches;
    return MP_OK;
}

// ------------------------------------------------------------------ models --------------------
extern "C++" {
namespace {
// t32: the [M,S,A] transitions, range-checked and narrowed by the caller
int load_table_common(mp_ctx *ctx, int32_t M, int32_t S, int32_t A, const std::vector<int32_t> &t32, const double *reward,
                      const uint8_t *terminal, int32_t done_on_next, int32_t max_steps, mp_model **out)
{
    const size_t n = (size_t)M * S * A;
    MP_HIP(hipSetDevice(ctx->device));
    mp_model *m = new mp_model();
    m->ctx = ctx; m->mode = MP_MODE_DETERMINISTIC; m->M = M; m->S = S; m->A = A; m->Sc = S; m->NB = 1; m->Sb = S;
    m->done_on_next = done_on_next ? 1 : 0; m->max_steps = max_steps > 0 ? max_steps : 0;
    auto bail = [&](int rc) { mp_model_free(m); return rc; };
    if (hipMalloc(&m->T, n * sizeof(int32_t)) != hipSuccess || hipMalloc(&m->R, n * sizeof(double)) != hipSuccess ||
        hipMalloc(&m->rec, (size_t)S * A * sizeof(Rec)) != hipSuccess)
        return bail(fail(MP_ERR_ALLOC, "mp_model_load_table: hipMalloc failed"));
    if (terminal && hipMalloc(&m->term, (size_t)S) != hipSuccess) return bail(fail(MP_ERR_ALLOC, "hipMalloc failed"));
    if (hipMemcpy(m->T, t32.data(), n * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->R, reward, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        (terminal && hipMemcpy(m->term, terminal, (size_t)S, hipMemcpyHostToDevice) != hipSuccess))
        return bail(fail(MP_ERR_HIP, "mp_model_load_table: upload failed"));
    const long sa = (long)S * A;
    hipLaunchKernelGGL(pack_records, dim3((unsigned)((sa + 255) / 256)), dim3(256), 0, ctx->stream, S, A, m->T, m->R,
                       m->term, (const uint8_t *)nullptr, m->rec);
    if (S < 32768) {
        if (hipMalloc(&m->t16, (((size_t)sa * 2 + 15) & ~(size_t)15) + 16) != hipSuccess)
            return bail(fail(MP_ERR_ALLOC, "mp_model_load_table: hipMalloc failed"));
        hipLaunchKernelGGL(pack_t16, dim3((unsigned)((sa + 255) / 256)), dim3(256), 0, ctx->stream, S, A, m->T, m->term,
                           m->t16);
        // the LDS-resident form of model 0 (uct.hip ENV_TABLE_LDSR): rewards as 8-bit indices into the table of their
        // distinct values (by bit pattern: -0.0 and 0.0, or two NaNs, stay what they are) -- when there are at most 256
        m->rmap = new std::unordered_map<uint64_t, int>();
        std::unordered_map<uint64_t, int> &seen = *m->rmap;
        std::vector<uint8_t> idx(((size_t)sa + 15) & ~(size_t)15, 0);
        std::vector<double> dict(256, 0.0);
        bool fits = true;
        for (long i = 0; i < sa && fits; ++i) {
            uint64_t bits;
            memcpy(&bits, &reward[i], sizeof(bits));
            auto it = seen.find(bits);
            if (it == seen.end()) {
                if (seen.size() == 256) { fits = false; break; }
                it = seen.emplace(bits, (int)seen.size()).first;
                dict[it->second] = reward[i];
            }
            idx[i] = (uint8_t)it->second;
        }
        if (fits) {
            if (hipMalloc(&m->r8, idx.size()) != hipSuccess || hipMalloc(&m->rdict, 256 * sizeof(double)) != hipSuccess)
                return bail(fail(MP_ERR_ALLOC, "mp_model_load_table: hipMalloc failed"));
            if (hipMemcpy(m->r8, idx.data(), idx.size(), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(m->rdict, dict.data(), 256 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
                return bail(fail(MP_ERR_HIP, "mp_model_load_table: upload failed"));
            m->n_rdict = (int)seen.size();
        } else {
            delete m->rmap;
            m->rmap = nullptr;
        }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail(fail(MP_ERR_HIP, "pack_records failed"));
    *out = m;
    return MP_OK;
}

// pinned staging block / device scratch block of a model's table updates, grown on demand
int update_blocks(mp_model *m, size_t stage_bytes, size_t dev_bytes)
{
    mp_ctx *ctx = m->ctx;
    if (!m->upd_done) MP_HIP(hipEventCreateWithFlags(&m->upd_done, hipEventDisableTiming));
    if (m->upd_pending) { // the previous update's copies still read the staging block
        MP_HIP(hipEventSynchronize(m->upd_done));
        m->upd_pending = false;
    }
    if (stage_bytes > m->upd_stage_cap) {
        if (m->upd_stage) MP_HIP(hipHostFree(m->upd_stage));
        m->upd_stage = nullptr; m->upd_stage_cap = 0;
        const size_t want = stage_bytes + stage_bytes / 4 + 4096;
        if (hipHostMalloc(&m->upd_stage, want, hipHostMallocDefault) != hipSuccess)
            return fail(MP_ERR_ALLOC, "model update: hipHostMalloc(%zu) failed", want);
        m->upd_stage_cap = want;
    }
    if (dev_bytes > m->upd_dev_cap) {
        if (m->upd_dev) {
            MP_HIP(hipStreamSynchronize(ctx->stream)); // (enqueued kernels of the previous update may still read it)
            MP_HIP(hipFree(m->upd_dev));
        }
        m->upd_dev = nullptr; m->upd_dev_cap = 0;
        const size_t want = dev_bytes + dev_bytes / 4 + 4096;
        if (hipMalloc(&m->upd_dev, want) != hipSuccess) return fail(MP_ERR_ALLOC, "model update: hipMalloc(%zu) failed", want);
        m->upd_dev_cap = want;
    }
    return MP_OK;
}

// index of reward r in the model's table of distinct rewards, extending it (host mirror + device copy, stream-ordered);
// -1: the table is full -- the model loses its LDS-resident form (the record-gather kernels serve it)
int reward_index(mp_model *m, double r, bool *grew)
{
    uint64_t bits;
    memcpy(&bits, &r, sizeof(bits));
    auto it = m->rmap->find(bits);
    if (it != m->rmap->end()) return it->second;
    if (m->rmap->size() >= 256) return -1;
    const int k = (int)m->rmap->size();
    m->rmap->emplace(bits, k);
    *grew = true;
    return k;
}

void drop_compact_rewards(mp_model *m)
{
    // (the arrays stay allocated -- enqueued kernels may still read them -- but no later launch sees them: the planners
    // pick the LDS-resident form by model->r8)
    if (m->r8) m->dead_blocks.push_back(m->r8);
    m->r8 = nullptr;
    m->n_rdict = 0;
    delete m->rmap;
    m->rmap = nullptr;
}
} // namespace
} // extern "C++"

int mp_model_load_table(mp_ctx *ctx, int32_t M, int32_t S, int32_t A, const int64_t *transition,
                        const double *reward, const uint8_t *terminal, int32_t done_on_next, int32_t max_steps,
                        mp_model **out)
{
    if (!ctx || !out || !transition || !reward) return fail(MP_ERR_ARG, "mp_model_load_table: NULL argument");
    if (M < 1 || S < 1 || A < 1) return fail(MP_ERR_ARG, "mp_model_load_table: bad shape M=%d S=%d A=%d", M, S, A);
    const size_t n = (size_t)M * S * A;
    std::vector<int32_t> t32(n);
    for (size_t i = 0; i < n; ++i) {
        const int64_t v = transition[i];
        if (v < 0 || v >= S) return fail(MP_ERR_ARG, "mp_model_load_table: transition[%zu] = %lld outside [0, %d)", i, (long long)v, S);
        t32[i] = (int32_t)v;
    }
    return load_table_common(ctx, M, S, A, t32, reward, terminal, done_on_next, max_steps, out);
}

int mp_model_load_table_batch(mp_ctx *ctx, int32_t N, int32_t S, int32_t A, const int64_t *transition, const double *reward,
                              const uint8_t *terminal, int32_t done_on_next, int32_t max_steps, mp_model **out)
{
    if (!ctx || !out || !transition || !reward) return fail(MP_ERR_ARG, "mp_model_load_table_batch: NULL argument");
    if (N < 1 || S < 1 || A < 1) return fail(MP_ERR_ARG, "mp_model_load_table_batch: bad shape N=%d S=%d A=%d", N, S, A);
    if ((int64_t)N * S * A >= ((int64_t)1 << 31))
        return fail(MP_ERR_ARG, "mp_model_load_table_batch: N * S * A = %lld does not fit 31 bits", (long long)N * S * A);
    const size_t n = (size_t)N * S * A, sa = (size_t)S * A;
    std::vector<int32_t> t32(n);
    for (size_t i = 0; i < n; ++i) {
        const int64_t v = transition[i];
        if (v < 0 || v >= S)
            return fail(MP_ERR_ARG, "mp_model_load_table_batch: transition[%zu] = %lld outside [0, %d)", i, (long long)v, S);
        t32[i] = (int32_t)(i / sa) * S + (int32_t)v; // global state of the MDP's own block
    }
    MP_TRY(load_table_common(ctx, 1, N * S, A, t32, reward, terminal, done_on_next, max_steps, out));
    (*out)->NB = N;
    (*out)->Sb = S;
    return MP_OK;
}

int mp_model_batch_info(const mp_model *m, int32_t *N, int32_t *S_each)
{
    if (!m) return fail(MP_ERR_ARG, "model is NULL");
    if (N) *N = m->NB;
    if (S_each) *S_each = m->NB > 1 || m->Sb > 0 ? m->Sb : m->S;
    return MP_OK;
}

int mp_model_update_tables(mp_model *m, int32_t first, int32_t count, const int64_t *transition, const double *reward,
                           const uint8_t *terminal)
{
    if (!m || !transition || !reward) return fail(MP_ERR_ARG, "mp_model_update_tables: NULL argument");
    if (m->mode != MP_MODE_DETERMINISTIC || !m->rec || m->M != 1 || m->rec_all)
        return fail(MP_ERR_MODE, "mp_model_update_tables: deterministic table models (M = 1) only");
    const int Sb = m->Sb > 0 ? m->Sb : m->S, A = m->A;
    if (first < 0 || count < 1 || first + count > m->NB)
        return fail(MP_ERR_ARG, "mp_model_update_tables: MDPs [%d, %d) of %d", first, first + count, m->NB);
    if ((terminal != nullptr) != (m->term != nullptr))
        return fail(MP_ERR_ARG, "mp_model_update_tables: terminal flags must be given iff the model was loaded with them");
    mp_ctx *ctx = m->ctx;
    MP_HIP(hipSetDevice(ctx->device));
    const size_t ns = (size_t)count * Sb, n = ns * A;
    const bool compact = m->r8 && m->rmap && m->n_rdict > 0;
    // staging layout: T int32 [n] | R double [n] | term uint8 [ns] | r8 uint8 [n] | rdict double [256]
    const size_t off_r = (n * 4 + 15) & ~(size_t)15, off_term = off_r + n * 8, off_r8 = (off_term + ns + 15) & ~(size_t)15,
                 off_dict = (off_r8 + n + 15) & ~(size_t)15, total = off_dict + 256 * 8;
    MP_TRY(update_blocks(m, total, 0));
    char *st = static_cast<char *>(m->upd_stage);
    int32_t *t32 = reinterpret_cast<int32_t *>(st);
    const size_t sa = (size_t)Sb * A;
    // (MDP by MDP: no division per element, and the range test folded into one unsigned compare accumulated over the block --
    // 4096 episodes x 600 pairs per lock-step of a per-episode evaluation go through here)
    for (int32_t b = 0; b < count; ++b) {
        const int64_t *src = transition + (size_t)b * sa;
        int32_t *dst = t32 + (size_t)b * sa;
        const int32_t base = (first + b) * Sb;
        uint64_t bad = 0;
        for (size_t j = 0; j < sa; ++j) {
            const uint64_t v = (uint64_t)src[j];
            bad |= v >= (uint64_t)Sb ? 1u : 0u;
            dst[j] = base + (int32_t)v;
        }
        if (bad)
            for (size_t j = 0; j < sa; ++j)
                if (src[j] < 0 || src[j] >= Sb)
                    return fail(MP_ERR_ARG, "mp_model_update_tables: transition[%zu] = %lld outside [0, %d)", (size_t)b * sa + j,
                                (long long)src[j], Sb);
    }
    memcpy(st + off_r, reward, n * 8);
    if (terminal) memcpy(st + off_term, terminal, ns);
    bool keep_compact = compact, grew = false;
    if (compact) {
        uint8_t *idx = reinterpret_cast<uint8_t *>(st + off_r8);
        for (size_t i = 0; i < n && keep_compact; ++i) {
            const int k = reward_index(m, reward[i], &grew);
            if (k < 0) keep_compact = false;
            else idx[i] = (uint8_t)k;
        }
    }
    hipStream_t s = ctx->stream;
    const size_t g0 = (size_t)first * Sb; // first global state
    MP_HIP(hipMemcpyAsync(m->T + g0 * A, t32, n * 4, hipMemcpyHostToDevice, s));
    MP_HIP(hipMemcpyAsync(m->R + g0 * A, st + off_r, n * 8, hipMemcpyHostToDevice, s));
    if (terminal) MP_HIP(hipMemcpyAsync(m->term + g0, st + off_term, ns, hipMemcpyHostToDevice, s));
    if (compact && keep_compact) {
        MP_HIP(hipMemcpyAsync(m->r8 + g0 * A, st + off_r8, n, hipMemcpyHostToDevice, s));
        if (grew) {
            double *dict = reinterpret_cast<double *>(st + off_dict);
            for (const auto &kv : *m->rmap) memcpy(&dict[kv.second], &kv.first, 8);
            MP_HIP(hipMemcpyAsync(m->rdict, dict, m->rmap->size() * 8, hipMemcpyHostToDevice, s));
            m->n_rdict = (int)m->rmap->size();
        }
    } else if (compact) {
        drop_compact_rewards(m);
    }
    hipLaunchKernelGGL(pack_records_range, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (int)g0, (int)ns, A, m->T, m->R,
                       m->term, (const uint8_t *)m->avail, m->rec, m->t16);
    MP_HIP(hipGetLastError());
    MP_HIP(hipEventRecord(m->upd_done, s));
    m->upd_pending = true;
    m->serial = mp::next_model_serial(); // policies fused from the old records no longer belong to this model
    return MP_OK;
}

int mp_model_update_rows(mp_model *m, int32_t n_rows, const int32_t *rows, const int64_t *transition, const double *reward,
                         const uint8_t *terminal)
{
    if (!m || !rows || !transition || !reward) return fail(MP_ERR_ARG, "mp_model_update_rows: NULL argument");
    if (m->mode != MP_MODE_DETERMINISTIC || !m->rec || m->M != 1 || m->rec_all)
        return fail(MP_ERR_MODE, "mp_model_update_rows: deterministic table models (M = 1) only");
    if (n_rows < 1) return fail(MP_ERR_ARG, "mp_model_update_rows: n_rows = %d", n_rows);
    if (terminal && !m->term) return fail(MP_ERR_ARG, "mp_model_update_rows: the model was loaded without terminal flags");
    const int Sb = m->Sb > 0 ? m->Sb : m->S, A = m->A;
    mp_ctx *ctx = m->ctx;
    MP_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)n_rows * A;
    const bool compact = m->r8 && m->rmap && m->n_rdict > 0;
    // staging = device scratch layout: rows int32 [n_rows] | T int32 [n] | R double [n] | term uint8 [n_rows] | r8 uint8 [n] | rdict [256]
    const size_t off_t = ((size_t)n_rows * 4 + 15) & ~(size_t)15, off_r = (off_t + n * 4 + 15) & ~(size_t)15, off_term = off_r + n * 8,
                 off_r8 = (off_term + n_rows + 15) & ~(size_t)15, off_dict = (off_r8 + n + 15) & ~(size_t)15, total = off_dict + 256 * 8;
    MP_TRY(update_blocks(m, total, total));
    char *st = static_cast<char *>(m->upd_stage);
    int32_t *hrows = reinterpret_cast<int32_t *>(st), *t32 = reinterpret_cast<int32_t *>(st + off_t);
    std::vector<int32_t> owners; // MDPs that own a listed row
    for (int k = 0; k < n_rows; ++k) {
        const int32_t g = rows[k];
        if (g < 0 || g >= m->S) return fail(MP_ERR_ARG, "mp_model_update_rows: rows[%d] = %d outside [0, %d)", k, g, m->S);
        hrows[k] = g;
        const int32_t b = g / Sb;
        if (owners.empty() || owners.back() != b) owners.push_back(b);
        for (int a = 0; a < A; ++a) {
            const int64_t v = transition[(size_t)k * A + a];
            if (v < 0 || v >= Sb)
                return fail(MP_ERR_ARG, "mp_model_update_rows: transition[%d, %d] = %lld outside [0, %d)", k, a, (long long)v, Sb);
            t32[(size_t)k * A + a] = b * Sb + (int32_t)v;
        }
    }
    memcpy(st + off_r, reward, n * 8);
    if (terminal) memcpy(st + off_term, terminal, (size_t)n_rows);
    bool keep_compact = compact, grew = false;
    if (compact) {
        uint8_t *idx = reinterpret_cast<uint8_t *>(st + off_r8);
        for (size_t i = 0; i < n && keep_compact; ++i) {
            const int k = reward_index(m, reward[i], &grew);
            if (k < 0) keep_compact = false;
            else idx[i] = (uint8_t)k;
        }
    }
    if (compact && keep_compact && grew) {
        double *dict = reinterpret_cast<double *>(st + off_dict);
        for (const auto &kv : *m->rmap) memcpy(&dict[kv.second], &kv.first, 8);
    }
    hipStream_t s = ctx->stream;
    char *dv = static_cast<char *>(m->upd_dev);
    MP_HIP(hipMemcpyAsync(dv, st, total, hipMemcpyHostToDevice, s));
    if (compact && keep_compact && grew) {
        MP_HIP(hipMemcpyAsync(m->rdict, dv + off_dict, m->rmap->size() * 8, hipMemcpyDeviceToDevice, s));
        m->n_rdict = (int)m->rmap->size();
    }
    if (compact && !keep_compact) drop_compact_rewards(m);
    const int32_t *drows = reinterpret_cast<const int32_t *>(dv);
    hipLaunchKernelGGL(scatter_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n_rows, A, drows,
                       reinterpret_cast<const int32_t *>(dv + off_t), reinterpret_cast<const double *>(dv + off_r),
                       terminal ? reinterpret_cast<const uint8_t *>(dv + off_term) : (const uint8_t *)nullptr,
                       compact && keep_compact ? reinterpret_cast<const uint8_t *>(dv + off_r8) : (const uint8_t *)nullptr, m->T, m->R,
                       m->term, compact && keep_compact ? m->r8 : (uint8_t *)nullptr);
    if (!terminal) {
        hipLaunchKernelGGL(pack_records_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n_rows, A, drows, m->T, m->R, m->term,
                           (const uint8_t *)m->avail, m->rec, m->t16);
    } else {
        // a record carries terminal[next]: every record of the owning MDPs follows (one launch per MDP; many owners: all)
        std::sort(owners.begin(), owners.end());
        owners.erase(std::unique(owners.begin(), owners.end()), owners.end());
        if (owners.size() > 64) {
            const long sa = (long)m->S * A;
            hipLaunchKernelGGL(pack_records_range, dim3((unsigned)((sa + 255) / 256)), dim3(256), 0, s, 0, m->S, A, m->T, m->R, m->term,
                               (const uint8_t *)m->avail, m->rec, m->t16);
        } else {
            const long sa = (long)Sb * A;
            for (int32_t b : owners)
                hipLaunchKernelGGL(pack_records_range, dim3((unsigned)((sa + 255) / 256)), dim3(256), 0, s, b * Sb, Sb, A, m->T, m->R,
                                   m->term, (const uint8_t *)m->avail, m->rec, m->t16);
        }
    }
    MP_HIP(hipGetLastError());
    MP_HIP(hipEventRecord(m->upd_done, s));
    m->upd_pending = true;
    m->serial = mp::next_model_serial();
    return MP_OK;
}

int mp_model_load_joint(mp_ctx *ctx, int32_t M, int32_t S, int32_t A, const int64_t *transition, const double *reward,
                        const uint8_t *terminal, int32_t done_on_next, mp_model **out)
{
    if (!ctx || !out || !transition || !reward) return fail(MP_ERR_ARG, "mp_model_load_joint: NULL argument");
    // tables, value-iteration view and model 0's records as for any table model (terminal flags of model 0)
    MP_TRY(mp_model_load_table(ctx, M, S, A, transition, reward, terminal, done_on_next, 0, out));
    mp_model *m = *out;
    auto bail = [&](int rc) { mp_model_free(m); *out = nullptr; return rc; };
    const long sa = (long)S * A;
    if (hipMalloc(&m->rec_all, (size_t)M * sa * sizeof(Rec)) != hipSuccess) return bail(fail(MP_ERR_ALLOC, "mp_model_load_joint: hipMalloc failed"));
    if (terminal) {
        if (hipMalloc(&m->term_all, (size_t)M * S) != hipSuccess) return bail(fail(MP_ERR_ALLOC, "mp_model_load_joint: hipMalloc failed"));
        if (hipMemcpy(m->term_all, terminal, (size_t)M * S, hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(MP_ERR_HIP, "mp_model_load_joint: upload failed"));
    }
    for (int k = 0; k < M; ++k)
        hipLaunchKernelGGL(pack_records, dim3((unsigned)((sa + 255) / 256)), dim3(256), 0, ctx->stream, S, A, m->T + k * sa,
                           m->R + k * sa, m->term_all ? (const uint8_t *)(m->term_all + (size_t)k * S) : (const uint8_t *)nullptr,
                           (const uint8_t *)nullptr, m->rec_all + k * sa);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail(fail(MP_ERR_HIP, "mp_model_load_joint: pack_records failed"));
    return MP_OK;
}

int mp_model_set_available(mp_model *m, const uint8_t *available)
{
    if (!m || !available) return fail(MP_ERR_ARG, "mp_model_set_available: NULL argument");
    if (m->mode != MP_MODE_DETERMINISTIC || !m->rec) return fail(MP_ERR_MODE, "mp_model_set_available: deterministic table models only");
    mp_ctx *ctx = m->ctx;
    const int S = m->S, A = m->A;
    for (int s = 0; s < S; ++s) {
        bool any = false;
        for (int a = 0; a < A; ++a) any |= available[(size_t)s * A + a] != 0;
        if (!any) return fail(MP_ERR_ARG, "mp_model_set_available: state %d has no available action", s);
    }
    MP_HIP(hipSetDevice(ctx->device));
    MP_HIP(hipStreamSynchronize(ctx->stream));
    if (!m->avail && hipMalloc(&m->avail, (size_t)S * A) != hipSuccess) return fail(MP_ERR_ALLOC, "mp_model_set_available: hipMalloc failed");
    MP_HIP(hipMemcpy(m->avail, available, (size_t)S * A, hipMemcpyHostToDevice));
    const long sa = (long)S * A;
    hipLaunchKernelGGL(pack_records, dim3((unsigned)((sa + 255) / 256)), dim3(256), 0, ctx->stream, S, A, m->T, m->R, m->term,
                       (const uint8_t *)m->avail, m->rec);
    MP_HIP(hipStreamSynchronize(ctx->stream));
    m->masked = true;
    m->serial = mp::next_model_serial(); // policies fused from the old records no longer belong to this model
    return MP_OK;
}

int mp_model_set_available_joint(mp_model *m, const uint8_t *available)
{
    if (!m || !available) return fail(MP_ERR_ARG, "mp_model_set_available_joint: NULL argument");
    if (m->mode != MP_MODE_DETERMINISTIC || !m->rec_all) return fail(MP_ERR_MODE, "mp_model_set_available_joint: joint models only");
    mp_ctx *ctx = m->ctx;
    const int S = m->S, A = m->A, M = m->M;
    for (long sm = 0; sm < (long)M * S; ++sm) {
        bool any = false;
        for (int a = 0; a < A; ++a) any |= available[(size_t)sm * A + a] != 0;
        if (!any) return fail(MP_ERR_ARG, "mp_model_set_available_joint: model %ld state %ld has no available action", sm / S, sm % S);
    }
    MP_HIP(hipSetDevice(ctx->device));
    MP_HIP(hipStreamSynchronize(ctx->stream));
    uint8_t *d = nullptr;
    if (hipMalloc(&d, (size_t)M * S * A) != hipSuccess) return fail(MP_ERR_ALLOC, "mp_model_set_available_joint: hipMalloc failed");
    if (hipMemcpy(d, available, (size_t)M * S * A, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return fail(MP_ERR_HIP, "mp_model_set_available_joint: upload failed");
    }
    const long sa = (long)S * A;
    for (int k = 0; k < M; ++k)
        hipLaunchKernelGGL(pack_records, dim3((unsigned)((sa + 255) / 256)), dim3(256), 0, ctx->stream, S, A, m->T + k * sa,
                           m->R + k * sa, m->term_all ? (const uint8_t *)(m->term_all + (size_t)k * S) : (const uint8_t *)nullptr,
                           (const uint8_t *)(d + (size_t)k * sa), m->rec_all + k * sa);
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(MP_ERR_HIP, "mp_model_set_available_joint: pack_records failed");
    m->masked = true;
    return MP_OK;
}

int mp_model_load_dense(mp_ctx *ctx, int32_t M, int32_t S, int32_t A, const double *transition,
                        const double *reward, const uint8_t *terminal, int32_t mem, mp_model **out)
{
    return mp_model_load_dense_rows(ctx, M, S, A, S, transition, reward, terminal, mem, out);
}

int mp_model_load_dense_rows(mp_ctx *ctx, int32_t M, int32_t S, int32_t A, int32_t S_cols, const double *transition,
                             const double *reward, const uint8_t *terminal, int32_t mem, mp_model **out)
{
    if (!ctx || !out || !transition || !reward) return fail(MP_ERR_ARG, "mp_model_load_dense: NULL argument");
    if (M < 1 || S < 1 || A < 1 || S_cols < 1) return fail(MP_ERR_ARG, "mp_model_load_dense: bad shape");
    MP_HIP(hipSetDevice(ctx->device));
    mp_model *m = new mp_model();
    m->ctx = ctx; m->mode = MP_MODE_STOCHASTIC; m->M = M; m->S = S; m->A = A; m->Sc = S_cols;
    auto bail = [&](int rc) { mp_model_free(m); return rc; };
    const size_t nr = (size_t)M * S * A, np = nr * S_cols;
    if (mem == MP_MEM_DEVICE) {
        m->P = transition; m->R = const_cast<double *>(reward); m->term = const_cast<uint8_t *>(terminal);
        m->borrowed = true;
    } else {
        double *p = nullptr;
        if (hipMalloc(&p, np * sizeof(double)) != hipSuccess) return bail(fail(MP_ERR_ALLOC, "hipMalloc(%zu) failed", np * 8));
        m->P = p;
        if (hipMalloc(&m->R, nr * sizeof(double)) != hipSuccess) return bail(fail(MP_ERR_ALLOC, "hipMalloc failed"));
        if (terminal && hipMalloc(&m->term, (size_t)S) != hipSuccess) return bail(fail(MP_ERR_ALLOC, "hipMalloc failed"));
        if (hipMemcpy(p, transition, np * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(m->R, reward, nr * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            (terminal && hipMemcpy(m->term, terminal, (size_t)S, hipMemcpyHostToDevice) != hipSuccess))
            return bail(fail(MP_ERR_HIP, "mp_model_load_dense: upload failed"));
    }
    *out = m;
    return MP_OK;
}

int mp_model_load_sparse(mp_ctx *ctx, int32_t S, int32_t A, int32_t B, const double *transition,
                         const int64_t *next, const double *reward, const uint8_t *terminal, mp_model **out)
{
    if (!ctx || !out || !transition || !reward || !next) return fail(MP_ERR_ARG, "mp_model_load_sparse: NULL argument");
    if (S < 1 || A < 1 || B < 1) return fail(MP_ERR_ARG, "mp_model_load_sparse: bad shape");
    const size_t n = (size_t)S * A * B;
    std::vector<int32_t> n32(n);
    for (size_t i = 0; i < n; ++i) {
        if (next[i] < 0 || next[i] >= S) return fail(MP_ERR_ARG, "mp_model_load_sparse: next[%zu] out of range", i);
        n32[i] = (int32_t)next[i];
    }
    MP_HIP(hipSetDevice(ctx->device));
    mp_model *m = new mp_model();
    m->ctx = ctx; m->mode = MP_MODE_SPARSE; m->M = 1; m->S = S; m->A = A; m->B = B; m->Sc = S;
    auto bail = [&](int rc) { mp_model_free(m); return rc; };
    double *p = nullptr;
    if (hipMalloc(&p, n * sizeof(double)) != hipSuccess) return bail(fail(MP_ERR_ALLOC, "hipMalloc failed"));
    m->P = p;
    if (hipMalloc(&m->NXT, n * sizeof(int32_t)) != hipSuccess || hipMalloc(&m->R, (size_t)S * A * sizeof(double)) != hipSuccess)
        return bail(fail(MP_ERR_ALLOC, "hipMalloc failed"));
    if (terminal && hipMalloc(&m->term, (size_t)S) != hipSuccess) return bail(fail(MP_ERR_ALLOC, "hipMalloc failed"));
    if (hipMemcpy(p, transition, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->NXT, n32.data(), n * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->R, reward, (size_t)S * A * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        (terminal && hipMemcpy(m->term, terminal, (size_t)S, hipMemcpyHostToDevice) != hipSuccess))
        return bail(fail(MP_ERR_HIP, "mp_model_load_sparse: upload failed"));
    *out = m;
    return MP_OK;
}

// Which restated form of glibc's small-argument sin / cos reproduces THIS host's libm (the reference's CartPole calls
// math.sin / math.cos): both forms against sin() / cos() on 40 000 angles over the range a pole can reach and beyond, plus the
// tiny-argument branches.  1 = the FMA-contracted form, 2 = every operation rounded, 0 = neither (another libm).
int mp_libm_sincos_variant(void)
{
    static int cached = -1;
    if (cached >= 0) return cached;
    bool ok[3] = {false, true, true};
    uint64_t lcg = 0x9E3779B97F4A7C15ULL;
    for (int i = 0; i < 40000; ++i) {
        lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
        const double u = (double)(lcg >> 11) * (1.0 / 9007199254740992.0);            // [0, 1)
        double x = (i % 4 == 0 ? 0.85 : (i % 4 == 1 ? 0.3 : (i % 4 == 2 ? 0.13 : 1e-6))) * (2.0 * u - 1.0);
        if (i % 997 == 0) x = ldexp(x, -30);
        const double s = sin(x), c = cos(x);
        if (libm_sin_small<true>(x) != s || libm_cos_small<true>(x) != c) ok[1] = false;
        if (libm_sin_small<false>(x) != s || libm_cos_small<false>(x) != c) ok[2] = false;
    }
    cached = ok[1] ? SINCOS_LIBM_FMA : (ok[2] ? SINCOS_LIBM_PLAIN : SINCOS_DEVICE);
    return cached;
}

int mp_libm_sincos(int32_t n, const double *x, int32_t variant, double *s, double *c)
{
    if (n < 0 || !x || !s || !c || variant < 0 || variant > 2) return fail(MP_ERR_ARG, "mp_libm_sincos: bad argument");
    for (int i = 0; i < n; ++i) libm_sincos(variant, x[i], &s[i], &c[i]);
    return MP_OK;
}

extern "C++" {
namespace mp {
__global__ void sincos_selftest_kernel(int n, int variant, const double *__restrict__ x, double *__restrict__ s, double *__restrict__ c)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // 3 / 4: the branch-free forms the CartPole rollouts evaluate (FMA-contracted / every operation rounded), valid for |x| < 0.855
    if (variant == 3) libm_sincos_small_flat<true>(x[i], &s[i], &c[i]);
    else if (variant == 4) libm_sincos_small_flat<false>(x[i], &s[i], &c[i]);
    else libm_sincos(variant, x[i], &s[i], &c[i]);
}
} // namespace mp
}

int mp_selftest_sincos(mp_ctx *ctx, int32_t n, const double *x, int32_t variant, double *s, double *c)
{
    if (!ctx || n < 1 || !x || !s || !c || variant < 0 || variant > 4) return fail(MP_ERR_ARG, "mp_selftest_sincos: bad argument");
    MP_HIP(hipSetDevice(ctx->device));
    double *d = nullptr;
    MP_HIP(hipMalloc(&d, (size_t)n * 3 * sizeof(double)));
    hipError_t e = hipMemcpyAsync(d, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(mp::sincos_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, variant, d, d + n, d + 2 * (size_t)n);
        e = hipMemcpyAsync(s, d + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(c, d + 2 * (size_t)n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    MP_HIP(e);
    MP_HIP(e2);
    return MP_OK;
}

int mp_model_load_cartpole(mp_ctx *ctx, const mp_cartpole_params *params, mp_model **out)
{
    if (!ctx || !out || !params) return fail(MP_ERR_ARG, "mp_model_load_cartpole: NULL argument");
    mp_model *m = new mp_model();
    m->ctx = ctx; m->mode = MP_MODE_CARTPOLE; m->M = 1; m->S = 0; m->A = 2;
    m->cp_sincos = mp_libm_sincos_variant();
    if (const char *e = getenv("MP_CARTPOLE_SINCOS")) // "device" / "fma" / "plain": measurement and test knob
        m->cp_sincos = !strcmp(e, "fma") ? SINCOS_LIBM_FMA : (!strcmp(e, "plain") ? SINCOS_LIBM_PLAIN : SINCOS_DEVICE);
    m->cp = *params;
    m->max_steps = params->max_steps;
    *out = m;
    return MP_OK;
}

int mp_model_free(mp_model *m)
{
    if (!m) return MP_OK;
    if (m->ctx) hipSetDevice(m->ctx->device);
    if (!m->borrowed) {
        if (m->P) hipFree(const_cast<double *>(m->P));
        if (m->R) hipFree(m->R);
        if (m->term) hipFree(m->term);
    }
    if (m->T) hipFree(m->T);
    if (m->rec) hipFree(m->rec);
    if (m->t16) hipFree(m->t16);
    if (m->r8) hipFree(m->r8);
    if (m->rdict) hipFree(m->rdict);
    if (m->sa_cost) hipFree(m->sa_cost);
    if (m->avail) hipFree(m->avail);
    if (m->rec_all) hipFree(m->rec_all);
    if (m->term_all) hipFree(m->term_all);
    if (m->NXT) hipFree(m->NXT);
    if (m->thr) hipFree(m->thr);
    if (m->srec) hipFree(m->srec);
    if (m->srec_rtab) hipFree(m->srec_rtab);
    if (m->upd_pending && m->upd_done) (void)hipEventSynchronize(m->upd_done);
    if (m->upd_stage) (void)hipHostFree(m->upd_stage);
    if (m->upd_dev) hipFree(m->upd_dev);
    if (m->upd_done) (void)hipEventDestroy(m->upd_done);
    for (void *p : m->dead_blocks) (void)hipFree(p);
    delete m->rmap;
    delete m;
    return MP_OK;
}

int mp_model_info(const mp_model *m, int32_t *mode, int32_t *M, int32_t *S, int32_t *A, int32_t *B)
{
    if (!m) return fail(MP_ERR_ARG, "model is NULL");
    if (mode) *mode = m->mode;
    if (M) *M = m->M;
    if (S) *S = m->S;
    if (A) *A = m->A;
    if (B) *B = m->B;
    return MP_OK;
}

// ------------------------------------------------------------------ OLOP.allocation -----------
// tree_search/olop.py:42-44
static int olop_horizon(int episodes, double gamma)
{
    const int h = (int)ceil(log((double)episodes) / (2.0 * log(1.0 / gamma)));
    return h > 1 ? h : 1;
}

// tree_search/olop.py:50-62 (ValueError -> MP_ERR_ARG)
int mp_olop_allocation(int32_t budget, double gamma, int32_t *episodes, int32_t *horizon)
{
    if (!episodes || !horizon) return fail(MP_ERR_ARG, "mp_olop_allocation: NULL output");
    for (int e = 1; e < budget; ++e) {
        if ((long long)e * olop_horizon(e, gamma) > budget) {
            const int ee = e - 1 > 1 ? e - 1 : 1;
            *episodes = ee;
            *horizon = olop_horizon(ee, gamma);
            return MP_OK;
        }
    }
    return fail(MP_ERR_ARG, "Could not split budget %d with gamma %g", budget, gamma);
}

} // extern "C"
