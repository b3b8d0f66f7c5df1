// api.hip -- context, transition-model upload and host helpers of libmi355plan.so.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "common.hpp"

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

// compact transitions for the LDS variant of the UCT kernel (S < 32768)
__global__ void pack_t16(int S, int A, const int32_t *__restrict__ T, const uint8_t *__restrict__ term,
                         uint16_t *__restrict__ t16)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)S * A) return;
    const int nx = T[i];
    t16[i] = (uint16_t)((uint32_t)nx | (term && term[nx] ? 0x8000u : 0u));
}

} // namespace mp

using namespace mp;

extern "C" {

const char *mp_last_error(void) { return g_err.c_str(); }
int mp_abi_version(void) { return MP_ABI_VERSION; }

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
    *out = ctx;
    return MP_OK;
}

int mp_ctx_destroy(mp_ctx *ctx)
{
    if (!ctx) return MP_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->vi_graph_exec) hipGraphExecDestroy((hipGraphExec_t)ctx->vi_graph_exec);
    for (auto &b : ctx->ws)
        if (b.p) hipFree(b.p);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
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

int mp_ctx_synchronize(mp_ctx *ctx)
{
    if (!ctx) return fail(MP_ERR_ARG, "ctx is NULL");
    MP_HIP(hipStreamSynchronize(ctx->stream));
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
    MP_HIP(hipSetDevice(ctx->device));
    mp_model *m = new mp_model();
    m->ctx = ctx; m->mode = MP_MODE_DETERMINISTIC; m->M = M; m->S = S; m->A = A; m->Sc = S;
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
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail(fail(MP_ERR_HIP, "pack_records failed"));
    *out = m;
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
    if (B > 128) return fail(MP_ERR_ARG, "mp_model_load_sparse: B=%d > 128 next-states per (s,a) not supported", B);
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

int mp_model_load_cartpole(mp_ctx *ctx, const mp_cartpole_params *params, mp_model **out)
{
    if (!ctx || !out || !params) return fail(MP_ERR_ARG, "mp_model_load_cartpole: NULL argument");
    mp_model *m = new mp_model();
    m->ctx = ctx; m->mode = MP_MODE_CARTPOLE; m->M = 1; m->S = 0; m->A = 2;
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
    if (m->avail) hipFree(m->avail);
    if (m->rec_all) hipFree(m->rec_all);
    if (m->term_all) hipFree(m->term_all);
    if (m->NXT) hipFree(m->NXT);
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
