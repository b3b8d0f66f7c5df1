// vi.hip -- Bellman sweeps of ValueIterationAgent / RobustValueIterationAgent on gfx950.
//
// Reference: dynamic_programming/value_iteration.py:37-73, robust_value_iteration.py:39-58.
//   Q_{k+1} = R + gamma * mask(next_v(max_a Q_k)),   stop when allclose(Q_k, Q_{k+1}) and return Q_k.
//
// Kernels
//   vi_det_sweep    deterministic tables, fused: keeps only V across sweeps (V_{k-1}, V_k -> V_{k+1}),
//                   recomputing Q_k and Q_{k+1} from the two V's for the allclose test, so a sweep
//                   moves 12*M*S*A + 17*S algorithmic bytes (T int32 + R f64 stream, V gather/write,
//                   terminal flags) and never materialises Q.  Bit-exact with numpy.
//   vi_dense_q      dense T[M,S,A,S]: the |S|x|A|x|S| contraction on the f64 matrix cores
//                   (v_mfma_f64_16x16x4_f64), 16 (s,a) rows per wave, V staged in LDS; HBM-bound
//                   (0.25 flop/B): 8*M*S^2*A bytes per sweep.
//   vi_sparse_q     sparse [S,A,B] model, numpy's pairwise add.reduce order restated (bit-exact).
//   vi_finish       V_{k+1} = max_a Q_{k+1}, allclose flag (dense / sparse paths).
// Convergence is decided on the device: sweep k raises notclose[k] if any element moved; sweep
// k+1 (already enqueued) turns into a no-op when notclose[k] stayed 0, freezing the iterates, so
// the whole solve is one asynchronous batch of launches with no host round trip.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <utility>
#include <vector>

#include "common.hpp"

namespace mp {

constexpr int kViPiece = 8192; // numpy.getbufsize(): what add.reduce hands its inner loop at a time (and the V window of VI_V_PIECES)

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double4_u __attribute__((ext_vector_type(4), aligned(8)));

__device__ __forceinline__ bool isclose_np(double a, double b, double rtol, double atol)
{
    // numpy.isclose(a, b): |a - b| <= atol + rtol * |b| for finite values, equality otherwise
    if (isfinite(a) && isfinite(b)) return fabs(a - b) <= atol + rtol * fabs(b);
    return a == b;
}

// the same predicate without a branch (both sides evaluated, selected): keeps fully unrolled loops straight-line code
__device__ __forceinline__ bool isclose_np_sel(double a, double b, double rtol, double atol)
{
    const bool fin = (int)isfinite(a) & (int)isfinite(b);
    const bool close = fabs(a - b) <= atol + rtol * fabs(b);
    const bool same = a == b;
    return fin ? close : same;
}

struct ViDetArgs {
    int M, S, A, robust, vform, k;
    const int32_t *T;
    const double *R;
    const uint8_t *term;
    double gamma, rtol, atol;
    const double *Vprev, *Vcur;
    double *Vnext;
    int32_t *notclose;
};

// Q[s,a] of the Bellman operator applied to V (value_iteration.py:51-63 deterministic branch;
// robust_value_iteration.py:46-58 with the min over models)
__device__ __forceinline__ double det_q(const ViDetArgs &p, const double *__restrict__ V, long sa, bool term_s)
{
    const long msa = (long)p.S * p.A;
    if (p.robust) {
        double best = 0.0;
        for (int m = 0; m < p.M; ++m) {
            const double nv = V[p.T[m * msa + sa]];
            const double qm = p.R[m * msa + sa] + p.gamma * nv;
            if (m == 0 || qm < best) best = qm;
        }
        return best;
    }
    const double nv = term_s ? 0.0 : V[p.T[sa]];
    return p.R[sa] + p.gamma * nv;
}

// AT > 0: |A| known at compile time -- every table load and V gather of a state is issued before the
// first use (two dependent round trips per sweep instead of 2*|A|); AT == 0: any |A|.
template <int AT>
__global__ __launch_bounds__(64) void vi_det_sweep(ViDetArgs p)
{
    if (p.k > 0 && p.notclose[p.k - 1] == 0) return; // converged at an earlier sweep: freeze
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= p.S) return;
    const bool term_s = (!p.robust && p.term) ? p.term[s] != 0 : false;
    bool nc = false;
    double vmax = 0.0;
    if (AT > 0 && !p.robust) {
        constexpr int AR = AT > 0 ? AT : 1;
        const long sa0 = (long)s * AR;
        int32_t t[AR];
        double r[AR], vc[AR], vp[AR];
#pragma unroll
        for (int a = 0; a < AR; ++a) { t[a] = p.T[sa0 + a]; r[a] = p.R[sa0 + a]; }
#pragma unroll
        for (int a = 0; a < AR; ++a) { vc[a] = p.Vcur[t[a]]; vp[a] = p.Vprev[t[a]]; }
#pragma unroll
        for (int a = 0; a < AR; ++a) {
            const double qn = r[a] + p.gamma * (term_s ? 0.0 : vc[a]);
            if (!p.vform) {
                const double qo = p.k == 0 ? 0.0 : r[a] + p.gamma * (term_s ? 0.0 : vp[a]);
                nc |= !isclose_np(qo, qn, p.rtol, p.atol);
            }
            if (a == 0 || qn > vmax) vmax = qn;
        }
    } else if (AT > 0) {
        // robust (robust_value_iteration.py:46-58): min over models, per model all |A| loads issued together
        constexpr int AR = AT > 0 ? AT : 1;
        const long msa = (long)p.S * AR, sa0 = (long)s * AR;
        double bn[AR], bo[AR];
        for (int m = 0; m < p.M; ++m) {
            int32_t t[AR];
            double r[AR], vc[AR], vp[AR];
#pragma unroll
            for (int a = 0; a < AR; ++a) { t[a] = p.T[m * msa + sa0 + a]; r[a] = p.R[m * msa + sa0 + a]; }
#pragma unroll
            for (int a = 0; a < AR; ++a) { vc[a] = p.Vcur[t[a]]; vp[a] = p.Vprev[t[a]]; }
#pragma unroll
            for (int a = 0; a < AR; ++a) {
                const double qm = r[a] + p.gamma * vc[a], qp = r[a] + p.gamma * vp[a];
                if (m == 0 || qm < bn[a]) bn[a] = qm;
                if (m == 0 || qp < bo[a]) bo[a] = qp;
            }
        }
#pragma unroll
        for (int a = 0; a < AR; ++a) {
            if (!p.vform) nc |= !isclose_np(p.k == 0 ? 0.0 : bo[a], bn[a], p.rtol, p.atol);
            if (a == 0 || bn[a] > vmax) vmax = bn[a];
        }
    } else {
        for (int a = 0; a < p.A; ++a) {
            const long sa = (long)s * p.A + a;
            const double qn = det_q(p, p.Vcur, sa, term_s);
            if (!p.vform) {
                const double qo = p.k == 0 ? 0.0 : det_q(p, p.Vprev, sa, term_s);
                nc |= !isclose_np(qo, qn, p.rtol, p.atol);
            }
            if (a == 0 || qn > vmax) vmax = qn;
        }
    }
    p.Vnext[s] = vmax;
    if (p.vform) nc = !isclose_np(p.Vcur[s], vmax, p.rtol, p.atol);
    if (nc) p.notclose[p.k] = 1;
}

// Small deterministic problems (12*M*S*A + 24*S bytes fit LDS, e.g. the reference's own S = 100 fixtures): the
// whole fixed-point iteration in ONE launch of one workgroup -- tables and the three V iterates live in LDS, a
// sweep is separated from the next by a workgroup barrier (~0.2 us) instead of a kernel boundary (~3 us), and the
// allclose early exit is a uniform branch.  Same arithmetic, same order, same returned iterate as the chained version.
struct ViSmallArgs {
    ViDetArgs d; // T / R / term in global memory, gamma, tolerances, M, S, A, robust, vform
    int iterations;
    double *Q_out, *V_out;
    int32_t *sweeps_out;
};

template <int AT>
__global__ __launch_bounds__(1024) void vi_det_small(ViSmallArgs q)
{
    extern __shared__ __attribute__((aligned(16))) double lds_v[];
    const ViDetArgs &p = q.d;
    const int S = p.S, A = AT > 0 ? AT : p.A, M = p.M, tid = threadIdx.x, nt = blockDim.x;
    const long msa = (long)S * A, n_sa = (long)M * msa;
    double *Vb = lds_v;                                        // [3][S]
    double *R = Vb + 3 * (long)S;                              // [M*S*A]
    int32_t *T = reinterpret_cast<int32_t *>(R + n_sa);        // [M*S*A]
    __shared__ int flag;
    for (long i = tid; i < 3L * S; i += nt) Vb[i] = 0.0;
    for (long i = tid; i < n_sa; i += nt) { R[i] = p.R[i]; T[i] = p.T[i]; }
    // this thread's states: s = tid, tid + nt, ... ; their terminal flags never change
    constexpr int kMaxOwn = 8;
    bool term_own[kMaxOwn];
#pragma unroll
    for (int i = 0; i < kMaxOwn; ++i) {
        const int s = tid + i * nt;
        term_own[i] = (!p.robust && p.term && s < S) ? p.term[s] != 0 : false;
    }
    __syncthreads();
    // Q[s, 0..A) of the Bellman operator applied to V, all loads of the row issued before the first use
    auto qrow = [&](const double *V, int s, bool term_s, double *out) {
        const long sa0 = (long)s * A;
        if (p.robust) {
            for (int a = 0; a < A; ++a) {
                double best = 0.0;
                for (int m = 0; m < M; ++m) {
                    const double qm = R[m * msa + sa0 + a] + p.gamma * V[T[m * msa + sa0 + a]];
                    if (m == 0 || qm < best) best = qm;
                }
                out[a] = best;
            }
        } else {
            for (int a = 0; a < A; ++a) out[a] = R[sa0 + a] + p.gamma * (term_s ? 0.0 : V[T[sa0 + a]]);
        }
    };
    constexpr int AR = AT > 0 ? AT : 64;
    int j = q.iterations, sweeps = q.iterations;
    for (int k = 0; k < q.iterations; ++k) {
        if (tid == 0) flag = 0;
        __syncthreads();
        const double *Vprev = Vb + (long)((k + 2) % 3) * S, *Vcur = Vb + (long)(k % 3) * S;
        double *Vnext = Vb + (long)((k + 1) % 3) * S;
        bool nc = false;
        int own = 0;
        for (int s = tid; s < S; s += nt, ++own) {
            const bool term_s = own < kMaxOwn ? term_own[own] : ((!p.robust && p.term) ? p.term[s] != 0 : false);
            double qn[AR], qo[AR];
            qrow(Vcur, s, term_s, qn);
            if (!p.vform && k > 0) qrow(Vprev, s, term_s, qo);
            double vmax = qn[0];
#pragma unroll
            for (int a = 0; a < AR; ++a) {
                if (a < A) {
                    if (!p.vform) nc |= !isclose_np(k == 0 ? 0.0 : qo[a], qn[a], p.rtol, p.atol);
                    if (a > 0 && qn[a] > vmax) vmax = qn[a];
                }
            }
            Vnext[s] = vmax;
            if (p.vform) nc |= !isclose_np(Vcur[s], vmax, p.rtol, p.atol);
        }
        if (nc) flag = 1;
        __syncthreads();
        if (flag == 0) { j = k; sweeps = k + 1; break; } // uniform: every thread reads the same flag
        __syncthreads();
    }
    if (tid == 0 && q.sweeps_out) *q.sweeps_out = sweeps;
    const double *Vj = Vb + (long)(j % 3) * S, *Vjm1 = Vb + (long)((j + 2) % 3) * S;
    for (int s = tid; s < S; s += nt) {
        if (q.V_out) q.V_out[s] = Vj[s];
        if (q.Q_out) {
            const bool term_s = (!p.robust && p.term) ? p.term[s] != 0 : false;
            double qj[AR];
            qrow(Vjm1, s, term_s, qj);
            for (int a = 0; a < A; ++a) q.Q_out[(long)s * A + a] = j == 0 ? 0.0 : qj[a];
        }
    }
}

// ------------------------------------------------------------------ persistent deterministic VI ---
// Deterministic problems too big for one workgroup's LDS but small enough that every state can own a thread of ONE
// co-resident grid (S <= 256 x 1024): the whole fixed-point iteration in ONE launch, synchronised by DATAFLOW.
//   * a thread keeps its state's T / R rows and its last Q rows in REGISTERS for the whole solve, so a sweep gathers
//     |A| V values, not 2|A| (the allclose test needs Q_k, which the chained launches recompute from V_{k-1});
//   * V lives in a ring of kRing slots of 16-byte entries = two self-validating 8-byte granules {sweep tag, half of the
//     f64}, written write-through and read L1-bypassing with relaxed agent-scope accesses (cdna_hip_programming.md
//     Guideline 16, form R2: "the data is the flag"): a thread starts sweep k as soon as ITS |A| successors carry tag
//     k -- no grid barrier, no fence, nothing but the data dependency of the algorithm on the critical path;
//   * the only global agreement the reference needs -- did ANY element move in sweep j (np.allclose) -- is taken off
//     that path: workgroups add {1 arrival, moved flag} to a per-sweep word and every workgroup reads the word of sweep
//     k - kLag at sweep k (by then complete in steady state: the wait is a lagged barrier that also bounds how far
//     workgroups drift apart: a writer in sweep k knows that everybody has left sweep k - 1 - kLag, so the slot it
//     overwrites -- V_{k+1-kRing} -- has no reader left with kRing = kLag + 2 slots).  Threads keep their last kLag + 1
//     Q rows, so the first sweep j with no movement -- found kLag sweeps late -- still returns the reference's Q_j.
struct ViPersistArgs {
    ViDetArgs d;      // T / R / term, gamma, tolerances, M, S, A, robust, vform
    int iterations, n_wg, block;
    unsigned long long *Vring; // [kRing][S][2] granules
    unsigned *sync;   // [0] timeout flag, [1 + k] arrivals (low 16 bits) + moved count (high bits) of sweep k
    double *Q_out, *V_out;
    int32_t *sweeps_out;
};

constexpr int kLag = 2, kRing = kLag + 2;
typedef __attribute__((address_space(1))) unsigned gu32_t;
typedef __attribute__((address_space(1))) unsigned long long gu64_t;
#define MP_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr unsigned kSpinLimit = 1u << 22;

template <int AT, int MT>
__global__ __launch_bounds__(1024) void vi_det_persist(ViPersistArgs q)
{
    const ViDetArgs &p = q.d;
    const int S = p.S, tid = threadIdx.x;
    const int s = blockIdx.x * blockDim.x + tid;
    const bool own = s < S;
    const long msa = (long)S * AT, sa0 = (long)(own ? s : 0) * AT;
    __shared__ unsigned bcast[4];
    int32_t t[MT][AT];
    double r[MT][AT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int a = 0; a < AT; ++a) { t[m][a] = p.T[m * msa + sa0 + a]; r[m][a] = p.R[m * msa + sa0 + a]; }
    const bool term_s = (!p.robust && p.term && own) ? p.term[s] != 0 : false;
    gu32_t *tmo = (gu32_t *)q.sync, *words = (gu32_t *)(q.sync + 1);
    gu64_t *ring = (gu64_t *)q.Vring;
    // Q_k, Q_{k-1}, ..., Q_{k-kLag} and the matching own V values; Q_0 = 0 (value_iteration.py:43)
    double qh[kLag + 1][AT], vh[kLag + 1];
#pragma unroll
    for (int h = 0; h <= kLag; ++h) {
        vh[h] = 0.0;
#pragma unroll
        for (int a = 0; a < AT; ++a) qh[h][a] = 0.0;
    }
    int stop = -1;      // first sweep j whose allclose test passed (returned iterate = Q_j)
    int hist = 0;       // which held row is the returned iterate
    bool failed = false;
    for (int k = 0; k < q.iterations; ++k) {
        // the verdict word of sweep k - kLag is normally complete by now: read it under the gather (off the chain)
        unsigned early = 0;
        if (tid == 0 && k >= kLag) early = __hip_atomic_load(words + (k - kLag), MP_RLX_AGENT);
        // ---- gather V_k at the successors: poll the granules until they carry tag k (V_0 = 0 needs none)
        double qn[AT];
        bool ok = true;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            double vc[AT];
#pragma unroll
            for (int a = 0; a < AT; ++a) vc[a] = 0.0;
            if (k > 0 && own) {
                // Both granules of a successor with ONE 16-byte L1-bypassing (sc1) load -- half the requests in the CU's
                // memory queue, where a hand-off's latency sits: 2.45-2.5 against 2.7 us per sweep at C2 (each half is
                // validated by its own tag, so a load torn between the two 8-byte stores is simply not ready yet).
                // Measured and dropped (round 4, profiles/r04_vi_persist_ab.txt): several probes in flight a fraction of a
                // round trip apart (3.4-4.9 us: the extra requests queue in front of the ones that matter), and the same solve
                // with single-wave workgroups and the arrival word sharded 64 ways (2.9 us at C2, 19 us at S = 50 000).
                const __amdgpu_buffer_rsrc_t rsrc =
                    __builtin_amdgcn_make_buffer_rsrc((void *)(q.Vring + (long)(k % kRing) * S * 2), 0, S * 16, 0x00020000);
                unsigned spins = 0;
                while (true) {
                    bool all = true;
#pragma unroll
                    for (int a = 0; a < AT; ++a) {
                        const uint4 g = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, t[m][a] * 16, 0, 16));
                        all &= g.y == (unsigned)k && g.w == (unsigned)k;
                        vc[a] = __hiloint2double((int)g.z, (int)g.x);
                    }
                    if (all) break;
                    if (++spins > kSpinLimit || __hip_atomic_load(tmo, MP_RLX_AGENT)) { ok = false; break; }
                    asm volatile("" ::: "memory"); // (the next probe reads memory again)
                }
            }
#pragma unroll
            for (int a = 0; a < AT; ++a) {
                const double qm = r[m][a] + p.gamma * (term_s ? 0.0 : vc[a]);
                if (m == 0 || qm < qn[a]) qn[a] = qm;       // robust_value_iteration.py:46-48 (MT = 1: plain VI)
            }
        }
        bool nc = false;
        double vmax = qn[0];
#pragma unroll
        for (int a = 0; a < AT; ++a) {
            if (!p.vform) nc |= !isclose_np(qh[0][a], qn[a], p.rtol, p.atol);
            if (a > 0 && qn[a] > vmax) vmax = qn[a];
        }
        if (p.vform) nc = !isclose_np(vh[0], vmax, p.rtol, p.atol);
        if (own && ok) { // publish V_{k+1}[s]: two granules tagged k + 1 (write-through)
            gu64_t *dst = ring + ((long)((k + 1) % kRing) * S + s) * 2;
            const unsigned long long tag = (unsigned long long)(unsigned)(k + 1) << 32;
            __hip_atomic_store(dst, tag | (unsigned)__double2loint(vmax), MP_RLX_AGENT);
            __hip_atomic_store(dst + 1, tag | (unsigned)__double2hiint(vmax), MP_RLX_AGENT);
        }
        // ---- per-workgroup: aggregate "moved" (bit 0) and "timed out" (bit 1), arrive for sweep k, publish the verdict
        // of sweep k - kLag to the other waves
        const int agg = __syncthreads_or((own && nc ? 1 : 0) | (ok ? 0 : 2));
        unsigned *bc = bcast + 2 * (k & 1);                 // double-buffered: one barrier per sweep
        if (tid == 0) {
            unsigned bad = (agg & 2) ? 1u : 0u, verdict = 1u;
            if (!bad) __hip_atomic_fetch_add(words + k, 1u + ((agg & 1) ? 0x10000u : 0u), MP_RLX_AGENT);
            if (!bad && k >= kLag) {
                unsigned spins = 0, w = early;
                while ((w & 0xffffu) < (unsigned)q.n_wg) {  // (steady state: complete already)
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kSpinLimit || __hip_atomic_load(tmo, MP_RLX_AGENT)) { bad = 1; break; }
                    w = __hip_atomic_load(words + (k - kLag), MP_RLX_AGENT);
                }
                verdict = w >> 16;
            }
            if (bad) __hip_atomic_store(tmo, 1u, MP_RLX_AGENT);
            bc[0] = bad; bc[1] = verdict;
        }
        __syncthreads();
        const unsigned bad = bc[0], verdict = bc[1];
        if (bad) { failed = true; break; }
        // rows held here: qh[h] = Q_{k-h}.  allclose(Q_j, Q_{j+1}) held at j = k - kLag: return Q_j = qh[kLag]
        if (k >= kLag && verdict == 0) { stop = k - kLag; hist = kLag; break; }
        // shift the history: qh[0] = Q_{k+1}
#pragma unroll
        for (int h = kLag; h > 0; --h) {
            vh[h] = vh[h - 1];
#pragma unroll
            for (int a = 0; a < AT; ++a) qh[h][a] = qh[h - 1][a];
        }
        vh[0] = vmax;
#pragma unroll
        for (int a = 0; a < AT; ++a) qh[0][a] = qn[a];
    }
    // ---- all `iterations` sweeps ran: rows qh[h] = Q_{N-h}.  The last kLag sweeps have not been judged yet; their
    // words are complete once every workgroup has arrived for them (bounded wait).
    if (!failed && stop < 0) {
        const int n_it = q.iterations;
        if (tid == 0) {
            unsigned bad = 0, first = 0xffffffffu;
            for (int j = max(0, n_it - kLag); j < n_it && !bad; ++j) {
                unsigned spins = 0, w;
                while (((w = __hip_atomic_load(words + j, MP_RLX_AGENT)) & 0xffffu) < (unsigned)q.n_wg) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kSpinLimit || __hip_atomic_load(tmo, MP_RLX_AGENT)) { bad = 1; break; }
                }
                if (!bad && (w >> 16) == 0 && first == 0xffffffffu) first = (unsigned)j;
            }
            if (bad) __hip_atomic_store(tmo, 1u, MP_RLX_AGENT);
            bcast[0] = bad; bcast[1] = first;
        }
        __syncthreads();
        if (bcast[0]) failed = true;
        else if (bcast[1] != 0xffffffffu) { stop = (int)bcast[1]; hist = n_it - stop; }
    }
    const int sweeps = failed ? -1 : (stop >= 0 ? stop + 1 : q.iterations);
    if (blockIdx.x == 0 && tid == 0 && q.sweeps_out) *q.sweeps_out = sweeps;
    if (own) {
        double vout = vh[0], qout[AT];
#pragma unroll
        for (int a = 0; a < AT; ++a) qout[a] = qh[0][a];
#pragma unroll
        for (int h = 1; h <= kLag; ++h)
            if (hist == h) {
                vout = vh[h];
#pragma unroll
                for (int a = 0; a < AT; ++a) qout[a] = qh[h][a];
            }
        if (q.V_out) q.V_out[s] = failed ? NAN : vout;
        if (q.Q_out)
#pragma unroll
            for (int a = 0; a < AT; ++a) q.Q_out[sa0 + a] = failed ? NAN : qout[a];
    }
}

template <int AT, int MT>
static void vi_persist_launch(const ViPersistArgs &q, hipStream_t st)
{
    hipLaunchKernelGGL((vi_det_persist<AT, MT>), dim3((unsigned)q.n_wg), dim3((unsigned)q.block), 0, st, q);
}

// The timeout word is the authority on "the grid was not co-resident" (a workgroup that finished before another one
// gave up has already written its own verdict): fold it into the caller's sweep count.
__global__ void vi_persist_status(const unsigned *sync, int32_t *sweeps_out)
{
    if (sync[0] && sweeps_out) *sweeps_out = -1;
}

// every (|A|, models) pair the persistent kernel is instantiated for; false = use the chained launches
static bool vi_persist_dispatch(const ViPersistArgs &q, int A, int M, hipStream_t st)
{
#define MP_VP(a, m) if (A == a && M == m) { vi_persist_launch<a, m>(q, st); return true; }
    MP_VP(2, 1) MP_VP(3, 1) MP_VP(4, 1) MP_VP(5, 1) MP_VP(6, 1) MP_VP(8, 1)
    MP_VP(2, 2) MP_VP(3, 2) MP_VP(4, 2) MP_VP(5, 2) MP_VP(6, 2) MP_VP(8, 2)
    MP_VP(2, 3) MP_VP(3, 3) MP_VP(4, 3) MP_VP(5, 3) MP_VP(2, 4) MP_VP(3, 4) MP_VP(4, 4)
#undef MP_VP
    return false;
}

template <int AT>
static int vi_small_launch(const ViSmallArgs &q, size_t lds, int threads, hipStream_t st)
{
    if (lds > 64 * 1024)
        MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vi_det_small<AT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    hipLaunchKernelGGL(vi_det_small<AT>, dim3(1), dim3((unsigned)threads), lds, st, q);
    return MP_OK;
}

static void vi_det_launch(const ViDetArgs &a, hipStream_t st)
{
    const dim3 grid((unsigned)((a.S + 63) / 64)), block(64);
    switch (a.A) {
    case 2: hipLaunchKernelGGL(vi_det_sweep<2>, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(vi_det_sweep<3>, grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL(vi_det_sweep<4>, grid, block, 0, st, a); break;
    case 5: hipLaunchKernelGGL(vi_det_sweep<5>, grid, block, 0, st, a); break;
    case 6: hipLaunchKernelGGL(vi_det_sweep<6>, grid, block, 0, st, a); break;
    case 8: hipLaunchKernelGGL(vi_det_sweep<8>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(vi_det_sweep<0>, grid, block, 0, st, a); break;
    }
}

// result[0] = sweeps executed, result[1] = j such that the returned iterate is Q_j / V_j
__global__ void vi_find_stop(int iterations, const int32_t *notclose, int32_t *result)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int j = iterations, sweeps = iterations;
    for (int k = 0; k < iterations; ++k)
        if (notclose[k] == 0) {
            j = k;
            sweeps = k + 1;
            break;
        }
    result[0] = sweeps;
    result[1] = j;
}

struct ViEmitArgs {
    ViDetArgs d;            // T/R/term/gamma, V pointers unused
    const double *Vbuf;     // 3 buffers of S doubles
    const int32_t *result;
    double *Q_out, *V_out;
    int32_t *sweeps_out;
};

// Q_j = 0 if j == 0 else Bellman(V_{j-1}); V_j straight from its buffer
__global__ __launch_bounds__(256) void vi_det_emit(ViEmitArgs e)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = e.result[1];
    if (s == 0 && e.sweeps_out) *e.sweeps_out = e.result[0];
    if (s >= e.d.S) return;
    if (e.V_out) e.V_out[s] = e.Vbuf[(long)(j % 3) * e.d.S + s];
    if (e.Q_out) {
        const bool term_s = (!e.d.robust && e.d.term) ? e.d.term[s] != 0 : false;
        const double *V = e.Vbuf + (long)((j + 2) % 3) * e.d.S; // V_{j-1}
        for (int a = 0; a < e.d.A; ++a) {
            const long sa = (long)s * e.d.A + a;
            e.Q_out[sa] = j == 0 ? 0.0 : det_q(e.d, V, sa, term_s);
        }
    }
}

// ------------------------------------------------------------------ generic (dense / sparse) ---
struct ViGenArgs {
    int M, S, A, B, robust, vform, k;
    int Sc; // columns of a dense model (= S unless the model holds a block of source-state rows)
    int nseg, seg_cols; // dense: column segments (blockIdx.y), columns per segment (a multiple of the V chunk); 1 = unsplit
    double *partial;    // dense, nseg > 1: per-segment dot products [M][nseg][S*A]
    const double *P;
    const int32_t *NXT;
    const double *R;
    const uint8_t *term;
    double gamma, rtol, atol;
    const double *Vcur; // V_k
    const double *Qcur; // Q_k
    double *Qnext;      // Q_{k+1}
    double *Vnext;      // V_{k+1}
    int32_t *notclose;
};

#ifndef MP_VI_DENSE_DEFAULT_EXACT
#define MP_VI_DENSE_DEFAULT_EXACT 1 // dense backups without MP_VI_DENSE / mp_vi_dense_mode: 1 = numpy's order (bit-exact, and measured
                                    // as fast as the matrix cores: HBM bounds both), 0 = matrix cores
#endif
#ifndef MP_DENSE_CHUNK
#define MP_DENSE_CHUNK 4096
#endif
#ifndef MP_DENSE_UNROLL
#define MP_DENSE_UNROLL 4
#endif
constexpr int kDenseChunk = MP_DENSE_CHUNK; // doubles of V staged in LDS per pass (32 KiB)
#ifndef MP_DENSE_SEG_CHUNKS
#define MP_DENSE_SEG_CHUNKS 2
#endif
constexpr int kDenseSegCols = MP_DENSE_SEG_CHUNKS * MP_DENSE_CHUNK; // columns per segment of the column split

// Rows of the (S*A) x S matrix T_m are contracted with V on v_mfma_f64_16x16x4_f64: the A operand
// of one MFMA is a 16-row x 4-column block of T, the B operand is V broadcast into all 16 columns,
// so every column of D holds the same 16 dot products.  Lane l feeds row (l & 15), k-slot (l >> 4);
// it loads 4 consecutive doubles (32 B) of its row per 16-column step, so the 4 lanes of a row
// cover one 128-B line and the k-slot <-> column map is (l >> 4) * 4 + t for MFMA t of the step
// (the same permutation is applied to V, which is all a dot product needs).
// Column split (nseg > 1): long rows (Sc > kDenseSegCols) are cut into segments of kDenseSegCols columns handled by
// different workgroups (blockIdx.y) -- a row block of a sharded model has few (s,a) rows (C5: 31 250 per rank = 489
// workgroups for 256 CUs), too few to keep enough loads in flight; the segments' partial dot products are summed in
// segment order by vi_dense_combine, so the result depends on Sc only (a row block reassembles the full backup bit for bit).
__global__ __launch_bounds__(256) void vi_dense_q(ViGenArgs p)
{
    if (p.k > 0 && p.notclose[p.k - 1] == 0) return;
    extern __shared__ __attribute__((aligned(16))) double vs[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long SA = (long)p.S * p.A;
    const long row0 = (long)blockIdx.x * 64 + wave * 16;
    const int i = lane & 15, q = lane >> 4;
    long row = row0 + i;
    if (row >= SA) row = SA - 1; // tail tile: load a valid row, discard the result
    const int seg = blockIdx.y;
    const int col_lo = seg * p.seg_cols, col_hi = min(p.Sc, col_lo + p.seg_cols);
    double best[4] = {0.0, 0.0, 0.0, 0.0};
    for (int m = 0; m < p.M; ++m) {
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
        const double *prow = p.P + ((long)m * SA + row) * p.Sc;
        for (int c0 = col_lo; c0 < col_hi; c0 += kDenseChunk) {
            const int ch = min(kDenseChunk, col_hi - c0);
            __syncthreads();
            for (int t = threadIdx.x; t < kDenseChunk; t += 256) vs[t] = t < ch ? p.Vcur[c0 + t] : 0.0;
            __syncthreads();
            const double *pr = prow + c0 + q * 4;
            const int full = ch & ~15;
            int cc = 0;
#pragma unroll MP_DENSE_UNROLL
            for (; cc < full; cc += 16) {
                const double4_u t4 = *reinterpret_cast<const double4_u *>(pr + cc);
                const double4_t b4 = *reinterpret_cast<const double4_t *>(vs + cc + q * 4);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(t4.x, b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(t4.y, b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(t4.z, b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(t4.w, b4.w, acc, 0, 0, 0);
            }
            if (cc < ch) { // ragged tail of the chunk: zero-fill beyond ch (V is zero-filled too)
                double tv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int col = cc + q * 4 + t;
                    tv[t] = col < ch ? pr[cc + t] : 0.0;
                }
                const double4_t b4 = *reinterpret_cast<const double4_t *>(vs + cc + q * 4);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[0], b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[1], b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[2], b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[3], b4.w, acc, 0, 0, 0);
            }
        }
        // f64 C/D layout: lane l, register r -> row (l >> 4) + 4 r, column l & 15
        const double accr[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long orow = row0 + q + 4 * r;
            if (orow < SA) {
                if (p.nseg > 1) {
                    if (i == 0) p.partial[((long)m * p.nseg + seg) * SA + orow] = accr[r];
                    continue;
                }
                const int s = (int)(orow / p.A);
                double nv = 0.0 + accr[r];
                if (!p.robust && p.term && p.term[s]) nv = 0.0;
                const double qm = p.R[(long)m * SA + orow] + p.gamma * nv;
                if (m == 0 || qm < best[r]) best[r] = qm;
            }
        }
    }
    if (i == 0 && p.nseg == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long orow = row0 + q + 4 * r;
            if (orow < SA) p.Qnext[orow] = best[r];
        }
    }
}

// nseg > 1: Q[row] = min_m (R_m + gamma * mask(sum over segments, in segment order))
__global__ __launch_bounds__(256) void vi_dense_combine(ViGenArgs p)
{
    if (p.k > 0 && p.notclose[p.k - 1] == 0) return;
    const long SA = (long)p.S * p.A;
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= SA) return;
    const int s = (int)(row / p.A);
    double best = 0.0;
    for (int m = 0; m < p.M; ++m) {
        double nv = 0.0;
        for (int g = 0; g < p.nseg; ++g) nv += p.partial[((long)m * p.nseg + g) * SA + row];
        if (!p.robust && p.term && p.term[s]) nv = 0.0;
        const double qm = p.R[(long)m * SA + row] + p.gamma * nv;
        if (m == 0 || qm < best) best = qm;
    }
    p.Qnext[row] = best;
}

// ---- the dense backup in NUMPY'S summation order: bit-exact with value_iteration.py:54-55 --------------------------------
// The reference computes (T * v.reshape(1, 1, S)).sum(axis=-1): every product rounded, then numpy's add.reduce over the
// contiguous axis -- pairwise summation: a row longer than 128 is halved (the left half rounded down to a multiple of 8)
// until the pieces ("leaves") hold at most 128 elements; a leaf is summed by eight strided accumulators r[j] += a[8 i + j],
// combined ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), its last len % 8 elements added one by one.  The matrix cores
// fuse and reorder (vi_dense_q: 1e-12); this kernel restates the order itself: ONE ROW PER WAVEFRONT, eight lanes per leaf
// (lane j of a group IS accumulator j: its loads are 8 elements apart, the eight lanes of a group read one 64-byte piece,
// and every byte of a line is used by two consecutive steps), eight leaves per pass, all loads of a pass in flight
// together; the group's three-level sum is a butterfly (IEEE addition commutes, so every lane ends with the reference's
// value); leaf results go to LDS and the recursion's additions are done level by level from a table the host derives
// from the row length -- including what the ufunc machinery adds: the reduction's iterator hands the inner loop at most
// numpy.getbufsize() = 8192 elements at a time, so a longer row is the running sum, from the identity 0., of the pairwise
// sums of its 8192-element pieces.  V is staged in LDS: all of it where it fits beside the tables (S <= ~14 000), else the
// piece being summed (VM below).  The kernel streams T once, as the matrix-core kernel does: it is bound by HBM, not by the
// order of its additions.
struct ViExactPlan {
    const int2 *leaves; // {offset, length} of the leaves of the recursion over Sc elements, left to right; leaf 0 = the identity
    const int2 *nodes;  // its additions by height: {left, right} result slots (leaf l = slot l, addition k = slot nleaf + k)
    const int *hoff;    // additions of height h + 1: [hoff[h], hoff[h + 1])
    const int *piece;   // leaves of the 8192-element piece c: [piece[c], piece[c + 1])
    int nleaf, nnode, nh, npiece;
};

enum { VI_V_GLOBAL = 0, VI_V_LDS = 1, VI_V_PIECES = 2 };

// VM: where the lanes read V from.  VI_V_LDS: all of it staged once per workgroup (it fits beside the tables up to ~14 000
// states).  VI_V_PIECES: longer rows -- the 8192-element piece the workgroup's waves are summing, staged between two
// barriers (the waves of a workgroup then walk their rows in step; read through L2 instead, every wave-load of T is
// matched by one of V that misses the 32 KB L1: 5.45 against 4.0 ms per sweep on a 25 GB row block).  VI_V_GLOBAL: no staging.
template <int NBT, int VM>
__global__ __launch_bounds__(512) void vi_dense_exact_q(ViGenArgs p, ViExactPlan pl)
{
    if (p.k > 0 && p.notclose[p.k - 1] == 0) return;
    extern __shared__ __attribute__((aligned(16))) double xs[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = (int)(blockDim.x >> 6);
    const int nslot = pl.nleaf + pl.nnode;
    // LDS: [leaves int2][nodes int2][hoff, piece int, padded to a double][result slots of every wave][V or its window]
    int2 *l_leaves = reinterpret_cast<int2 *>(xs);
    int2 *l_nodes = l_leaves + pl.nleaf;
    int *l_hoff = reinterpret_cast<int *>(l_nodes + pl.nnode);
    int *l_piece = l_hoff + pl.nh + 1;
    const int tab = nslot + (pl.nh + pl.npiece + 3) / 2;
    double *slots = xs + tab + (long)wave * nslot;
    double *vs = xs + tab + (long)nw * nslot;
    for (int t = threadIdx.x; t < pl.nleaf; t += blockDim.x) l_leaves[t] = pl.leaves[t];
    for (int t = threadIdx.x; t < pl.nnode; t += blockDim.x) l_nodes[t] = pl.nodes[t];
    for (int t = threadIdx.x; t <= pl.nh; t += blockDim.x) l_hoff[t] = pl.hoff[t];
    for (int t = threadIdx.x; t <= pl.npiece; t += blockDim.x) l_piece[t] = pl.piece[t];
    if (VM == VI_V_LDS)
        for (int t = threadIdx.x; t < p.Sc; t += blockDim.x) vs[t] = p.Vcur[t];
    __syncthreads();
    const long SA = (long)p.S * p.A;
    const int g = lane >> 3, j = lane & 7;
    constexpr int VPRE = VM == VI_V_PIECES ? kViPiece / 512 : 1; // window elements a thread prefetches (all of them from 512 threads on)
    double vpre[VPRE];
    if (VM == VI_V_PIECES) {
        const int nn = min(kViPiece, p.Sc);
#pragma unroll
        for (int i = 0; i < VPRE; ++i) {
            const int t = (int)threadIdx.x + i * (int)blockDim.x;
            vpre[i] = p.Vcur[t < nn ? t : 0];
        }
    }
    // A PASS = eight leaves of one (row, model): the unit of the loop below.  Passes are software-pipelined: the loads of pass
    // n + 1 -- of the same row, of its next model, or of the wave's next row -- are issued BEFORE pass n is summed, so a wave
    // keeps up to 2 x 16 loads of 512 bytes in flight and the reduction / combine phases hide under them (one pass at a time
    // left the memory system idle while a wave added: 0.74 of the HBM peak against the matrix-core kernel's 0.74-0.78).
    // (VI_V_PIECES: every wave of the workgroup walks the same sequence of passes -- the barriers of the window staging --
    // so a wave past the last row sums the last row again and does not store.)
    struct Pass {
        long row0;      // first row of the workgroup's row group (row = row0 + wave)
        int m, c, l0;   // model, 8192-element piece, first leaf of the pass
        int nb, base, off, len, l;
        bool valid, lvalid;
    };
    const long stride = (long)gridDim.x * nw;
    auto row_of = [&](const Pass &q) -> long {
        const long r = q.row0 + wave;
        return r < SA ? r : SA - 1;
    };
    auto alive = [&](long row0) -> bool { return VM == VI_V_PIECES ? row0 < SA : row0 + wave < SA; };
    auto issue = [&](Pass &q, double (&x)[NBT]) { // leaf of this lane's group + the loads of its eight accumulators' steps
        const int l_end = l_piece[q.c + 1];
        q.l = q.l0 + g;
        q.lvalid = q.l < l_end;
        const int2 lf = l_leaves[q.lvalid ? q.l : l_end - 1];
        q.off = lf.x;
        q.len = q.lvalid ? lf.y : 0;
        q.nb = q.len >> 3;                                      // full steps of the eight accumulators
        const int vbase = VM == VI_V_PIECES ? q.c * kViPiece : 0;
        q.base = q.nb > 0 ? q.off + j : vbase;                  // (a lane without a step re-reads an element that is there)
        const double *prow = p.P + ((long)q.m * SA + row_of(q)) * p.Sc;
#pragma unroll
        for (int i = 0; i < NBT; ++i) x[i] = prow[q.base + (i < q.nb ? 8 * i : 0)];
    };
    auto advance = [&](const Pass &q) -> Pass {
        Pass n = q;
        n.l0 = q.l0 + 8;
        if (n.l0 >= l_piece[q.c + 1]) {
            n.c = q.c + 1;
            if (n.c >= pl.npiece) {
                n.c = 0;
                n.m = q.m + 1;
                if (n.m >= p.M) { n.m = 0; n.row0 = q.row0 + stride; }
            }
            n.l0 = l_piece[n.c];
        }
        n.valid = alive(n.row0);
        return n;
    };
    double best = 0.0;
    auto compute = [&](const Pass &q, const double (&x)[NBT]) {
        const long row = row_of(q);
        const double *prow = p.P + ((long)q.m * SA + row) * p.Sc;
        const int vbase = VM == VI_V_PIECES ? q.c * kViPiece : 0;
        if (q.l0 == l_piece[q.c]) {                             // first pass of a piece
            if (q.c == 0 && lane == 0) slots[0] = 0.0;          // leaf 0: the identity the reduction starts from
            if (VM == VI_V_PIECES) {
                // the window was requested one piece ago (vpre, below): its round trip to L2 hides behind the passes of
                // the previous piece instead of standing between two barriers
                __syncthreads();                                // the previous window's readers are done
                const int cn = min(kViPiece, p.Sc - vbase);
#pragma unroll
                for (int i = 0; i < VPRE; ++i) {
                    const int t = (int)threadIdx.x + i * (int)blockDim.x;
                    if (t < cn) vs[t] = vpre[i];
                }
                for (int t = threadIdx.x + VPRE * blockDim.x; t < cn; t += blockDim.x) vs[t] = p.Vcur[vbase + t]; // (< 512 threads)
                __syncthreads();
                const int nbase = (q.c + 1 < pl.npiece ? q.c + 1 : 0) * kViPiece; // every (rows, model) walks the pieces in order
                const int nn = min(kViPiece, p.Sc - nbase);
#pragma unroll
                for (int i = 0; i < VPRE; ++i) {
                    const int t = (int)threadIdx.x + i * (int)blockDim.x;
                    vpre[i] = p.Vcur[nbase + (t < nn ? t : 0)];
                }
            }
        }
        double v[NBT];
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
            const int idx = q.base + (i < q.nb ? 8 * i : 0);
            v[i] = VM == VI_V_GLOBAL ? p.Vcur[idx] : vs[idx - vbase];
        }
        double r = q.nb > 0 ? x[0] * v[0] : 0.0;                // r[j] = a[j] (n < 8: res = 0.)
#pragma unroll
        for (int i = 1; i < NBT; ++i) {
            const double t = x[i] * v[i];
            r = i < q.nb ? r + t : r;
        }
        r += __shfl_xor(r, 1);                                  // r0 + r1 | r2 + r3 | r4 + r5 | r6 + r7
        r += __shfl_xor(r, 2);                                  // (r0 + r1) + (r2 + r3) | (r4 + r5) + (r6 + r7)
        r += __shfl_xor(r, 4);
        const int rem = q.len & 7;
        if (__any(rem != 0)) {                                  // only the last leaf of a row can have a remainder
            const int tail = q.off + 8 * q.nb;
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                const int idx = t < rem ? tail + t : vbase;
                const double pv = prow[idx] * (VM == VI_V_GLOBAL ? p.Vcur[idx] : vs[idx - vbase]);
                r = t < rem ? r + pv : r;
            }
        }
        if (j == 0 && q.lvalid) slots[q.l] = r;
        if (q.c == pl.npiece - 1 && q.l0 + 8 >= l_piece[pl.npiece]) { // last pass of this (row, model): the recursion's additions
            __builtin_amdgcn_wave_barrier();
            for (int h = 0; h < pl.nh; ++h) {
                for (int k = l_hoff[h] + lane; k < l_hoff[h + 1]; k += 64) {
                    const int2 nd = l_nodes[k];
                    slots[pl.nleaf + k] = slots[nd.x] + slots[nd.y];
                }
                __builtin_amdgcn_wave_barrier();
            }
            double nv = slots[nslot - 1];
            __builtin_amdgcn_wave_barrier();
            const int s = (int)(row / p.A);
            if (!p.robust && p.term && p.term[s]) nv = 0.0;
            const double qm = p.R[(long)q.m * SA + row] + p.gamma * nv;
            if (q.m == 0 || qm < best) best = qm;
            if (q.m == p.M - 1 && lane == 0 && q.row0 + wave < SA) p.Qnext[row] = best;
        }
    };
    Pass pa, pb;
    double xa[NBT], xb[NBT];
    pa.row0 = (long)blockIdx.x * nw; pa.m = 0; pa.c = 0; pa.l0 = l_piece[0];
    pa.valid = alive(pa.row0);
    pb = pa; pb.valid = false;
    if (pa.valid) issue(pa, xa);
    while (pa.valid) {
        pb = advance(pa);
        if (pb.valid) issue(pb, xb);
        compute(pa, xa);
        if (!pb.valid) break;
        pa = advance(pb);
        if (pa.valid) issue(pa, xa);
        compute(pb, xb);
    }
}

// the recursion of numpy's pairwise sum over n elements as tables (see vi_dense_exact_q)
static void vi_exact_plan_host(int n, std::vector<int> &leaves, std::vector<int> &nodes, std::vector<int> &hoff, std::vector<int> &piece,
                               int *nb_max)
{
    struct Add { int l, r, h; };
    std::vector<Add> adds;
    leaves.clear();
    // returns {slot, height}; leaves take slots 0.., additions are renumbered by height below
    struct Rec {
        std::vector<int> &lv; std::vector<Add> &ad;
        std::pair<int, int> go(int off, int len)
        {
            if (len <= 128) { lv.push_back(off); lv.push_back(len); return {(int)lv.size() / 2 - 1, 0}; }
            int n2 = len / 2;
            n2 -= n2 % 8;
            const auto a = go(off, n2), b = go(off + n2, len - n2);
            const int h = 1 + (a.second > b.second ? a.second : b.second);
            ad.push_back({a.first, b.first, h});
            return {-(int)ad.size(), h}; // additions as negative provisional ids
        }
    } rec{leaves, adds};
    // add.reduce hands its inner loop at most numpy.getbufsize() = 8192 elements at a time: a longer row is the running
    // sum, from the identity 0., of the pairwise sums of its 8192-element pieces.  The identity is leaf 0 (no elements).
    leaves.push_back(0); leaves.push_back(0);
    piece.clear();
    int acc = 0, acc_h = 0;
    for (int off = 0; off < n; off += kViPiece) {
        piece.push_back((int)leaves.size() / 2);
        const auto sum = rec.go(off, n - off < kViPiece ? n - off : kViPiece);
        acc_h = 1 + (acc_h > sum.second ? acc_h : sum.second);
        adds.push_back({acc, sum.first, acc_h});
        acc = -(int)adds.size();
    }
    piece.push_back((int)leaves.size() / 2);
    const int nleaf = (int)leaves.size() / 2, nnode = (int)adds.size();
    int nh = 0, nbm = 1;
    for (const Add &a : adds) nh = a.h > nh ? a.h : nh;
    for (int l = 0; l < nleaf; ++l) nbm = leaves[2 * l + 1] / 8 > nbm ? leaves[2 * l + 1] / 8 : nbm;
    // stable counting sort of the additions by height (1..nh); the slot of the addition at position k is nleaf + k, the
    // root -- the only addition of the greatest height -- comes last
    std::vector<int> start(nh + 2, 0);
    for (const Add &a : adds) ++start[a.h + 1];
    for (int h = 1; h <= nh + 1; ++h) start[h] += start[h - 1]; // start[h] = additions of height < h
    std::vector<int> cur(start), order(nnode), newid(nnode);
    for (int k = 0; k < nnode; ++k) { const int pos = cur[adds[k].h]++; order[pos] = k; newid[k] = pos; }
    auto slot = [&](int id) { return id >= 0 ? id : nleaf + newid[-id - 1]; };
    nodes.assign(2 * (size_t)nnode, 0);
    for (int pos = 0; pos < nnode; ++pos) {
        nodes[2 * pos] = slot(adds[order[pos]].l);
        nodes[2 * pos + 1] = slot(adds[order[pos]].r);
    }
    hoff.assign(nh + 1, 0);
    for (int h = 0; h <= nh; ++h) hoff[h] = start[h + 1]; // heights 1..nh -> [hoff[h - 1], hoff[h])
    *nb_max = nbm;
}

static bool vi_dense_exact_on(mp_ctx *ctx)
{
    if (ctx->vi_dense_exact < 0) {
        const char *e = getenv("MP_VI_DENSE");
        ctx->vi_dense_exact = (e && !strcmp(e, "exact")) ? 1 : ((e && !strcmp(e, "mfma")) ? 0 : MP_VI_DENSE_DEFAULT_EXACT);
    }
    return ctx->vi_dense_exact == 1;
}

template <int NBT, int VM>
static int vi_dense_exact_launch_vm(const ViGenArgs &a, const ViExactPlan &pl, unsigned grid, unsigned block, size_t lds, hipStream_t st)
{
    MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vi_dense_exact_q<NBT, VM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((vi_dense_exact_q<NBT, VM>), dim3(grid), dim3(block), lds, st, a, pl);
    return MP_OK;
}

template <int NBT>
static int vi_dense_exact_launch_nb(const ViGenArgs &a, const ViExactPlan &pl, int vm, unsigned grid, unsigned block, size_t lds,
                                    hipStream_t st)
{
    if (vm == VI_V_LDS) return vi_dense_exact_launch_vm<NBT, VI_V_LDS>(a, pl, grid, block, lds, st);
    if (vm == VI_V_PIECES) return vi_dense_exact_launch_vm<NBT, VI_V_PIECES>(a, pl, grid, block, lds, st);
    return vi_dense_exact_launch_vm<NBT, VI_V_GLOBAL>(a, pl, grid, block, lds, st);
}

static int vi_dense_exact_launch(mp_ctx *ctx, ViGenArgs &a, hipStream_t st, int *launches)
{
    if (ctx->vi_exact_cols != a.Sc) { // the summation tables of this row length (a few KB, uploaded once)
        std::vector<int> leaves, nodes, hoff, piece;
        int nbm = 1;
        vi_exact_plan_host(a.Sc, leaves, nodes, hoff, piece, &nbm);
        std::vector<int> all(leaves);
        all.insert(all.end(), nodes.begin(), nodes.end());
        all.insert(all.end(), hoff.begin(), hoff.end());
        all.insert(all.end(), piece.begin(), piece.end());
        int *d = nullptr;
        ctx->vi_exact_cols = 0;
        MP_TRY(ws_get(ctx, WS_VI5, all.size(), &d));
        MP_HIP(hipMemcpyAsync(d, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice, st));
        MP_HIP(hipStreamSynchronize(st)); // (`all` is a local)
        ctx->vi_exact_cols = a.Sc;
        ctx->vi_exact_nleaf = (int)leaves.size() / 2; ctx->vi_exact_nnode = (int)nodes.size() / 2;
        ctx->vi_exact_nh = (int)hoff.size() - 1; ctx->vi_exact_nb = nbm; ctx->vi_exact_npiece = (int)piece.size() - 1;
    }
    ViExactPlan pl;
    int *d = static_cast<int *>(ctx->ws[WS_VI5].p);
    pl.nleaf = ctx->vi_exact_nleaf; pl.nnode = ctx->vi_exact_nnode; pl.nh = ctx->vi_exact_nh; pl.npiece = ctx->vi_exact_npiece;
    pl.leaves = reinterpret_cast<const int2 *>(d);
    pl.nodes = reinterpret_cast<const int2 *>(d + 2 * pl.nleaf);
    pl.hoff = d + 2 * pl.nleaf + 2 * pl.nnode;
    pl.piece = pl.hoff + pl.nh + 1;
    const long SA = (long)a.S * a.A;
    const int nslot = pl.nleaf + pl.nnode;
    // Eight waves per workgroup: measured 0.669 ms per sweep at S = 10 000 against 0.700 with sixteen (50 000 rows over 2 048
    // waves leave a shorter tail than over 4 096) -- each wave keeps up to sixteen 512-byte loads in flight.
    int nw = 8;
    if (const char *e = getenv("MP_VI_EXACT_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 8) nw = v; } // (launch bounds: 512 threads)
    const size_t tab = (size_t)nslot + (size_t)(pl.nh + pl.npiece + 3) / 2;
    while (nw > 1 && (tab + (size_t)nw * nslot) * sizeof(double) > kLdsBytes / 2) nw >>= 1;
    const size_t fixed = (tab + (size_t)nw * nslot) * sizeof(double);
    if (fixed > kLdsBytes - 1024) return fail(MP_ERR_ARG, "vi: rows of %d columns need %zu bytes of LDS for the summation tables", a.Sc, fixed);
    int vm = VI_V_GLOBAL;
    if (fixed + (size_t)a.Sc * sizeof(double) <= kLdsBytes - 1024) vm = VI_V_LDS;
    else if (fixed + (size_t)kViPiece * sizeof(double) <= kLdsBytes - 1024) vm = VI_V_PIECES;
    if (const char *e = getenv("MP_VI_EXACT_V")) { // test / measurement knob: global | pieces (whole-V staging only where it fits)
        if (!strcmp(e, "global")) vm = VI_V_GLOBAL;
        else if (!strcmp(e, "pieces") && fixed + (size_t)kViPiece * sizeof(double) <= kLdsBytes - 1024) vm = VI_V_PIECES;
    }
    if (getenv("MP_VI_EXACT_NO_VLDS")) vm = VI_V_GLOBAL;
    const size_t vbytes = vm == VI_V_LDS ? (size_t)a.Sc * sizeof(double) : (vm == VI_V_PIECES ? (size_t)kViPiece * sizeof(double) : 0);
    const size_t lds = fixed + vbytes;
    const long groups = (SA + nw - 1) / nw;
    int wg_per_cu = (int)((kLdsBytes - 1024) / (lds > 0 ? lds : 1));
    if (wg_per_cu > 2048 / (64 * nw)) wg_per_cu = 2048 / (64 * nw);
    if (vm != VI_V_GLOBAL) wg_per_cu = 1; // one staged copy of V per CU
    if (wg_per_cu < 1) wg_per_cu = 1;
    const long cap = (long)ctx->prop.multiProcessorCount * wg_per_cu;
    const unsigned grid = (unsigned)(groups < cap ? groups : cap);
    const unsigned block = 64u * (unsigned)nw;
    const int nb = ctx->vi_exact_nb;
    int rc;
    if (nb <= 8) rc = vi_dense_exact_launch_nb<8>(a, pl, vm, grid, block, lds, st);
    else if (nb <= 10) rc = vi_dense_exact_launch_nb<10>(a, pl, vm, grid, block, lds, st);
    else if (nb <= 12) rc = vi_dense_exact_launch_nb<12>(a, pl, vm, grid, block, lds, st);
    else if (nb <= 14) rc = vi_dense_exact_launch_nb<14>(a, pl, vm, grid, block, lds, st);
    else rc = vi_dense_exact_launch_nb<16>(a, pl, vm, grid, block, lds, st);
    if (rc != MP_OK) return rc;
    if (launches) ++*launches;
    return MP_OK;
}

// launch the dense backup (split by column segments when the rows are long)
static int vi_dense_launch(mp_ctx *ctx, ViGenArgs &a, hipStream_t st, int *launches)
{
    if (vi_dense_exact_on(ctx)) return vi_dense_exact_launch(ctx, a, st, launches);
    const long SA = (long)a.S * a.A;
    a.seg_cols = kDenseSegCols;
    a.nseg = a.Sc > kDenseSegCols && !getenv("MP_DENSE_NO_SPLIT") ? (a.Sc + kDenseSegCols - 1) / kDenseSegCols : 1;
    if (a.nseg == 1) a.seg_cols = a.Sc;
    a.partial = nullptr;
    if (a.nseg > 1) MP_TRY(ws_get(ctx, WS_VI4, (size_t)a.M * a.nseg * SA, &a.partial));
    hipLaunchKernelGGL(vi_dense_q, dim3((unsigned)((SA + 63) / 64), (unsigned)a.nseg), dim3(256), kDenseChunk * sizeof(double), st, a);
    if (launches) ++*launches;
    if (a.nseg > 1) {
        hipLaunchKernelGGL(vi_dense_combine, dim3((unsigned)((SA + 255) / 256)), dim3(256), 0, st, a);
        if (launches) ++*launches;
    }
    return MP_OK;
}

// numpy add.reduce inner loop for n <= 128 contiguous doubles (pairwise summation base cases)
template <typename F>
__device__ __forceinline__ double np_sum_le128(int n, F elem)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += elem(i);
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = elem(j);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += elem(i + j);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += elem(i);
    return res;
}

// numpy's pairwise recursion above the 128-element blocks (a thread's own row: B next states), and add.reduce's running sum
// over the 8192-element pieces its iterator cuts a long row into (see vi_dense_exact_q)
template <typename F>
__device__ double np_pairwise(int off, int n, const F &elem)
{
    if (n <= 128) return np_sum_le128(n, [&](int i) { return elem(off + i); });
    int n2 = n / 2;
    n2 -= n2 % 8;
    const double a = np_pairwise(off, n2, elem);
    return a + np_pairwise(off + n2, n - n2, elem);
}

// value_iteration.py:56-59: (P * np.take(v, next)).sum(axis=-1)
template <bool BIG>
__global__ __launch_bounds__(256) void vi_sparse_q(ViGenArgs p)
{
    if (p.k > 0 && p.notclose[p.k - 1] == 0) return;
    const long sa = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (sa >= (long)p.S * p.A) return;
    const int s = (int)(sa / p.A);
    const double *pp = p.P + sa * p.B;
    const int32_t *nn = p.NXT + sa * p.B;
    const double *V = p.Vcur;
    auto elem = [&](int b) { return pp[b] * V[nn[b]]; };
    double nv = 0.0;
    if (BIG) { // more than 128 next states per (s, a): the recursion (device stack), piece by piece
        for (int off = 0; off < p.B; off += kViPiece) nv += np_pairwise(off, min(kViPiece, p.B - off), elem);
    } else {
        nv += np_sum_le128(p.B, elem);
    }
    if (p.term && p.term[s]) nv = 0.0;
    p.Qnext[sa] = p.R[sa] + p.gamma * nv;
}

__global__ __launch_bounds__(256) void vi_finish(ViGenArgs p)
{
    if (p.k > 0 && p.notclose[p.k - 1] == 0) return;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= p.S) return;
    bool nc = false;
    double vmax = 0.0;
    for (int a = 0; a < p.A; ++a) {
        const long sa = (long)s * p.A + a;
        const double qn = p.Qnext[sa];
        if (!p.vform) nc |= !isclose_np(p.Qcur[sa], qn, p.rtol, p.atol);
        if (a == 0 || qn > vmax) vmax = qn;
    }
    p.Vnext[s] = vmax;
    if (p.vform) nc = !isclose_np(p.Vcur[s], vmax, p.rtol, p.atol);
    if (nc) p.notclose[p.k] = 1;
}

struct ViGenEmitArgs {
    int S, A;
    const double *Qbuf; // 2 x S*A
    const double *Vbuf; // 2 x S
    const int32_t *result;
    double *Q_out, *V_out;
    int32_t *sweeps_out;
};

__global__ __launch_bounds__(256) void vi_gen_emit(ViGenEmitArgs e)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = e.result[1];
    const long SA = (long)e.S * e.A;
    if (i == 0 && e.sweeps_out) *e.sweeps_out = e.result[0];
    if (e.Q_out && i < SA) e.Q_out[i] = e.Qbuf[(long)(j & 1) * SA + i];
    if (e.V_out && i < e.S) e.V_out[i] = e.Vbuf[(long)(j & 1) * e.S + i];
}

// The persistent kernel needs the whole grid co-resident (at most one workgroup per CU is used, well inside what the
// hardware admits) and an instantiation for (|A|, models).  MP_VI_NO_PERSIST=1: chained launches.
// threads per workgroup: few waves per CU keep the consumer's memory queue short (the hand-off latency sits there),
// few workgroups keep the per-sweep arrival atomics (one word, ~12 ns each) off the throughput limit
static int vi_persist_block(int S)
{
    if (const char *e = getenv("MP_VI_PERSIST_BLOCK")) {
        const int v = atoi(e);
        if (v == 64 || v == 128 || v == 256 || v == 512 || v == 1024) return v;
    }
    (void)S;
    return 256;
}

static bool vi_persist_ok(mp_ctx *ctx, int S, int A, int M)
{
    if (getenv("MP_VI_NO_PERSIST")) return false;
    const int n_wg = (S + vi_persist_block(S) - 1) / vi_persist_block(S);
    if (n_wg > ctx->prop.multiProcessorCount) return false;
    // Measured on MI355X (tools/micro_vi_persist.py): 40 workgroups (S = 10 000) 2.6 us per sweep against
    // 3.0 us for the chained launches, and the early exit really ends the solve; 196 workgroups (S = 50 000, M = 2)
    // 5.4 us against 4.2 us -- the per-sweep arrivals serialise on one word and the hand-offs queue behind each other.
    if (n_wg > 64 && !getenv("MP_VI_PERSIST_BLOCK")) return false;
    const bool at = A == 2 || A == 3 || A == 4 || A == 5 || A == 6 || A == 8;
    if (!at || M < 1 || M > 4) return false;
    if (M == 3 && A > 5) return false;
    if (M == 4 && A > 4) return false;
    return true;
}

static int vi_run_impl(mp_ctx *ctx, mp_model *m, double gamma, int iterations, double rtol, double atol, int robust,
                       int vform, double *Q_out, double *V_out, int32_t *sweeps_out, int mem, bool allow_persist,
                       bool *persist_failed)
{
    if (!ctx || !m) return fail(MP_ERR_ARG, "vi: NULL ctx/model");
    if (iterations < 0) return fail(MP_ERR_ARG, "vi: iterations < 0");
    if (m->mode == MP_MODE_CARTPOLE) return fail(MP_ERR_MODE, "vi: the environment must be of type finite_mdp");
    if (robust && m->mode == MP_MODE_SPARSE) return fail(MP_ERR_MODE, "Unknown mode"); // robust_value_iteration.py:57-58
    if (m->Sc != m->S) return fail(MP_ERR_MODE, "vi: a row-block model can only be used with mp_vi_backup");
    if (m->NB > 1)
        return fail(MP_ERR_MODE, "vi: a batch model holds %d independent MDPs, each with its own convergence test: use mp_vi_solve_batch", m->NB);
    MP_HIP(hipSetDevice(ctx->device));
    const int S = m->S, A = m->A, M = robust ? m->M : 1;
    const long SA = (long)S * A;
    hipStream_t st = ctx->stream;

    int32_t *notclose = nullptr, *result = nullptr;
    MP_TRY(ws_get(ctx, WS_VI0, (size_t)iterations + 4, &notclose));
    result = notclose + iterations + 1;
    MP_HIP(hipMemsetAsync(notclose, 0, ((size_t)iterations + 4) * sizeof(int32_t), st));

    double *dQ = nullptr, *dV = nullptr;
    int32_t *dSw = nullptr;
    MP_TRY(stage_out_alloc(ctx, WS_IO0, Q_out, (size_t)SA, mem, &dQ));
    MP_TRY(stage_out_alloc(ctx, WS_IO1, V_out, (size_t)S, mem, &dV));
    MP_TRY(stage_out_alloc(ctx, WS_IO2, sweeps_out, 1, mem, &dSw));

    const unsigned gs = (unsigned)((S + 255) / 256);
    int launches = 0;
    bool persisted = false;
    unsigned *persist_sync = nullptr;
    const size_t small_lds = (size_t)3 * S * sizeof(double) + (size_t)M * SA * (sizeof(double) + sizeof(int32_t));
    if (m->mode == MP_MODE_DETERMINISTIC && small_lds <= kLdsBytes - 2048 && A <= 64 && !getenv("MP_VI_NO_SMALL")) {
        ViSmallArgs q;
        memset(&q, 0, sizeof(q));
        q.d.M = M; q.d.S = S; q.d.A = A; q.d.robust = robust; q.d.vform = vform;
        q.d.T = m->T; q.d.R = m->R; q.d.term = m->term; q.d.gamma = gamma; q.d.rtol = rtol; q.d.atol = atol;
        q.iterations = iterations; q.Q_out = dQ; q.V_out = dV; q.sweeps_out = dSw;
        MP_TRY(kernels_begin(ctx));
        const int threads = S >= 1024 ? 1024 : ((S + 63) / 64) * 64;
        switch (A) {
        case 2: MP_TRY(vi_small_launch<2>(q, small_lds, threads, st)); break;
        case 3: MP_TRY(vi_small_launch<3>(q, small_lds, threads, st)); break;
        case 4: MP_TRY(vi_small_launch<4>(q, small_lds, threads, st)); break;
        case 5: MP_TRY(vi_small_launch<5>(q, small_lds, threads, st)); break;
        case 6: MP_TRY(vi_small_launch<6>(q, small_lds, threads, st)); break;
        case 8: MP_TRY(vi_small_launch<8>(q, small_lds, threads, st)); break;
        default:
            if (A > 64) return fail(MP_ERR_ARG, "vi: |A| = %d > 64 not supported by the single-launch kernel", A);
            MP_TRY(vi_small_launch<0>(q, small_lds, threads, st));
            break;
        }
        MP_TRY(kernels_end(ctx, 1));
    } else if (m->mode == MP_MODE_DETERMINISTIC && allow_persist && vi_persist_ok(ctx, S, A, M)) {
        // one persistent launch: every state owns a thread of a co-resident grid (see vi_det_persist)
        persisted = true;
        ViPersistArgs q;
        memset(&q, 0, sizeof(q));
        q.d.M = M; q.d.S = S; q.d.A = A; q.d.robust = robust; q.d.vform = vform;
        q.d.T = m->T; q.d.R = m->R; q.d.term = m->term; q.d.gamma = gamma; q.d.rtol = rtol; q.d.atol = atol;
        q.iterations = iterations;
        q.block = vi_persist_block(S);
        q.n_wg = (S + q.block - 1) / q.block;
        q.Q_out = dQ; q.V_out = dV; q.sweeps_out = dSw;
        MP_TRY(ws_get(ctx, WS_VI1, (size_t)kRing * S * 2, &q.Vring));
        MP_TRY(ws_get(ctx, WS_VI3, (size_t)iterations + 4, &q.sync));
        // Guideline 16: every polled word is re-initialised by every call (tags 0 never match a sweep >= 1)
        MP_HIP(hipMemsetAsync(q.Vring, 0, (size_t)kRing * S * 2 * sizeof(unsigned long long), st));
        MP_HIP(hipMemsetAsync(q.sync, 0, ((size_t)iterations + 4) * sizeof(unsigned), st));
        // test hook: a raised timeout word is what a grid that is not co-resident ends in (after its bounded spins)
        if (getenv("MP_VI_PERSIST_INJECT_TIMEOUT")) MP_HIP(hipMemsetAsync(q.sync, 1, 1, st));
        persist_sync = q.sync;
        MP_TRY(kernels_begin(ctx));
        if (!vi_persist_dispatch(q, A, M, st)) return fail(MP_ERR_ARG, "vi: no persistent kernel for |A| = %d, M = %d", A, M);
        MP_TRY(kernels_end(ctx, 1));
        hipLaunchKernelGGL(vi_persist_status, dim3(1), dim3(1), 0, st, q.sync, dSw);
    } else if (m->mode == MP_MODE_DETERMINISTIC) {
        double *Vb = nullptr;
        MP_TRY(ws_get(ctx, WS_VI1, (size_t)3 * S, &Vb));
        MP_HIP(hipMemsetAsync(Vb, 0, (size_t)3 * S * sizeof(double), st));
        ViDetArgs a;
        a.M = M; a.S = S; a.A = A; a.robust = robust; a.vform = vform;
        a.T = m->T; a.R = m->R; a.term = m->term; a.gamma = gamma; a.rtol = rtol; a.atol = atol;
        a.notclose = notclose;
        // The sweep chain is launch-bound (a sweep moves < 1 MB): capture it once into a hipGraph and replay
        // it, so the host enqueues one graph instead of `iterations` kernels.  The graph is keyed by every
        // value baked into its kernel arguments.
        ViGraphKey key;
        memset(&key, 0, sizeof(key));
        key.model = m; key.T = m->T; key.R = m->R; key.term = m->term; key.Vb = Vb; key.notclose = notclose;
        key.iterations = iterations; key.M = M; key.S = S; key.A = A; key.robust = robust; key.vform = vform;
        key.gamma = gamma; key.rtol = rtol; key.atol = atol;
        const bool use_graph = iterations >= 8 && !getenv("MP_VI_NO_GRAPH");
        MP_TRY(kernels_begin(ctx));
        if (use_graph) {
            if (!ctx->vi_graph_exec || memcmp(&ctx->vi_graph_key, &key, sizeof(key)) != 0) {
                if (ctx->vi_graph_exec) {
                    MP_HIP(hipGraphExecDestroy((hipGraphExec_t)ctx->vi_graph_exec));
                    ctx->vi_graph_exec = nullptr;
                }
                hipGraph_t graph = nullptr;
                MP_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                for (int k = 0; k < iterations; ++k) {
                    a.k = k;
                    a.Vprev = Vb + (long)((k + 2) % 3) * S;
                    a.Vcur = Vb + (long)(k % 3) * S;
                    a.Vnext = Vb + (long)((k + 1) % 3) * S;
                    vi_det_launch(a, st);
                }
                MP_HIP(hipStreamEndCapture(st, &graph));
                hipGraphExec_t exec = nullptr;
                MP_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                MP_HIP(hipGraphDestroy(graph));
                ctx->vi_graph_exec = exec;
                memcpy(&ctx->vi_graph_key, &key, sizeof(key));
            }
            MP_HIP(hipGraphLaunch((hipGraphExec_t)ctx->vi_graph_exec, st));
            launches = iterations;
        } else {
            for (int k = 0; k < iterations; ++k) {
                a.k = k;
                a.Vprev = Vb + (long)((k + 2) % 3) * S;
                a.Vcur = Vb + (long)(k % 3) * S;
                a.Vnext = Vb + (long)((k + 1) % 3) * S;
                vi_det_launch(a, st);
                ++launches;
            }
        }
        MP_TRY(kernels_end(ctx, launches));
        a.k = 0;
        hipLaunchKernelGGL(vi_find_stop, dim3(1), dim3(64), 0, st, iterations, notclose, result);
        ViEmitArgs e;
        e.d = a; e.Vbuf = Vb; e.result = result; e.Q_out = dQ; e.V_out = dV; e.sweeps_out = dSw;
        hipLaunchKernelGGL(vi_det_emit, dim3(gs), dim3(256), 0, st, e);
    } else {
        double *Qb = nullptr, *Vb = nullptr;
        MP_TRY(ws_get(ctx, WS_VI1, (size_t)2 * S, &Vb));
        MP_TRY(ws_get(ctx, WS_VI2, (size_t)2 * SA, &Qb));
        MP_HIP(hipMemsetAsync(Vb, 0, (size_t)2 * S * sizeof(double), st));
        MP_HIP(hipMemsetAsync(Qb, 0, (size_t)2 * SA * sizeof(double), st));
        ViGenArgs a;
        a.M = M; a.S = S; a.A = A; a.B = m->B; a.robust = robust; a.vform = vform; a.Sc = S;
        a.P = m->P; a.NXT = m->NXT; a.R = m->R; a.term = m->term; a.gamma = gamma; a.rtol = rtol; a.atol = atol;
        a.notclose = notclose;
        const unsigned gq_sparse = (unsigned)((SA + 255) / 256);
        MP_TRY(kernels_begin(ctx));
        for (int k = 0; k < iterations; ++k) {
            a.k = k;
            a.Vcur = Vb + (long)(k & 1) * S;
            a.Vnext = Vb + (long)((k + 1) & 1) * S;
            a.Qcur = Qb + (long)(k & 1) * SA;
            a.Qnext = Qb + (long)((k + 1) & 1) * SA;
            if (m->mode == MP_MODE_STOCHASTIC) {
                MP_TRY(vi_dense_launch(ctx, a, st, &launches));
            } else {
                if (a.B > 128) hipLaunchKernelGGL(vi_sparse_q<true>, dim3(gq_sparse), dim3(256), 0, st, a);
                else hipLaunchKernelGGL(vi_sparse_q<false>, dim3(gq_sparse), dim3(256), 0, st, a);
                ++launches;
            }
            hipLaunchKernelGGL(vi_finish, dim3(gs), dim3(256), 0, st, a);
            ++launches;
        }
        MP_TRY(kernels_end(ctx, launches));
        hipLaunchKernelGGL(vi_find_stop, dim3(1), dim3(64), 0, st, iterations, notclose, result);
        ViGenEmitArgs e;
        e.S = S; e.A = A; e.Qbuf = Qb; e.Vbuf = Vb; e.result = result; e.Q_out = dQ; e.V_out = dV; e.sweeps_out = dSw;
        hipLaunchKernelGGL(vi_gen_emit, dim3((unsigned)((SA + 255) / 256)), dim3(256), 0, st, e);
    }
    MP_HIP(hipGetLastError());
    MP_TRY(stage_out_copy(ctx, Q_out, dQ, (size_t)SA, mem));
    MP_TRY(stage_out_copy(ctx, V_out, dV, (size_t)S, mem));
    MP_TRY(stage_out_copy(ctx, sweeps_out, dSw, 1, mem));
    if (mem == MP_MEM_HOST) {
        unsigned timed_out = 0;
        if (persisted) MP_HIP(hipMemcpyAsync(&timed_out, persist_sync, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        MP_HIP(hipStreamSynchronize(st));
        if (persisted && timed_out && persist_failed) *persist_failed = true;
    }
    return MP_OK;
}

// vi_det_persist spins across workgroups, i.e. it needs its whole grid co-resident.  The grid is at most 64
// workgroups on a 256-CU device, but a GPU shared with another stream / process / rank can still leave part of it
// waiting; every workgroup then gives up after its bounded spins, raises the timeout word and the outputs are NaN with
// sweeps = -1.  With host arrays the call has synchronised anyway: it reads the word and, when raised, solves again on
// the chained launches (which need no residency) -- the caller never sees the failure.  With device arrays the call is
// asynchronous: `sweeps_out` = -1 (and NaN in Q / V) is the documented report (mi355plan.h), which the Python wrappers
// turn into an exception.
static int vi_run(mp_ctx *ctx, mp_model *m, double gamma, int iterations, double rtol, double atol, int robust,
                  int vform, double *Q_out, double *V_out, int32_t *sweeps_out, int mem)
{
    bool persist_failed = false;
    MP_TRY(vi_run_impl(ctx, m, gamma, iterations, rtol, atol, robust, vform, Q_out, V_out, sweeps_out, mem, true,
                       &persist_failed));
    if (persist_failed)
        return vi_run_impl(ctx, m, gamma, iterations, rtol, atol, robust, vform, Q_out, V_out, sweeps_out, mem, false,
                           nullptr);
    return MP_OK;
}

// ------------------------------------------------------------------ batched deterministic VI ---
// N independent MDPs (a batch model: mp_model_load_table_batch), ONE WORKGROUP PER MDP, one launch: what N
// ValueIterationAgent objects compute (value_iteration.py:42-73 each, one agent per process in trainer/evaluation.py:139-194).
// Every MDP runs to its own allclose exit -- a uniform branch of its workgroup -- and returns its own iterate.  The tables of
// the batch hold GLOBAL next states (b * Sb + s'); a workgroup subtracts its base.
struct ViBatchArgs {
    int N, Sb, A, iterations;
    const int32_t *T;     // [N*Sb*A] global next states
    const double *R;      // [N*Sb*A]
    const uint8_t *term;  // [N*Sb] or nullptr
    double gamma, rtol, atol;
    double *Q_out;        // [N*Sb*A]
    int32_t *sweeps_out;  // [N]
    double *Vglobal;      // VGLOBAL form: [N][3][Sb]
    int only_failed;      // VGLOBAL form as the cluster form's fallback: solve only the MDPs whose sweeps_out says -1
};

// REGISTER form (Sb <= OWN * BLOCK): a thread keeps the rows of its OWN states -- transitions, rewards and the last Q row,
// which is what the allclose test compares with -- in registers for the whole solve; V is double-buffered in LDS; one
// barrier per sweep (it also carries the "did anything move" vote).
template <int AT, int OWN, int BLOCK>
__global__ __launch_bounds__(BLOCK) void vi_det_batch_reg(ViBatchArgs p)
{
    extern __shared__ __attribute__((aligned(16))) double lds_v[];
    const int b = blockIdx.x, tid = threadIdx.x, S = p.Sb;
    const long base = (long)b * S;
    double *V0 = lds_v, *V1 = lds_v + S;
    int32_t t[OWN][AT];
    double r[OWN][AT], qp[OWN][AT];
    bool term_s[OWN], own[OWN];
#pragma unroll
    for (int i = 0; i < OWN; ++i) {
        const int s = tid + i * BLOCK;
        own[i] = s < S;
        const long sa0 = (base + (own[i] ? s : 0)) * AT;
        term_s[i] = (p.term && own[i]) ? p.term[base + s] != 0 : false;
#pragma unroll
        for (int a = 0; a < AT; ++a) {
            t[i][a] = p.T[sa0 + a] - (int32_t)base;
            r[i][a] = p.R[sa0 + a];
            qp[i][a] = 0.0;                                  // Q_0 = 0 (value_iteration.py:43)
        }
        if (own[i]) V0[s] = 0.0;
    }
    __syncthreads();
    int sweeps = p.iterations;
    for (int k = 0; k < p.iterations; ++k) {
        const double *Vcur = (k & 1) ? V1 : V0;
        double *Vnext = (k & 1) ? V0 : V1;
        bool nc = false;
        double qn[OWN][AT];
#pragma unroll
        for (int i = 0; i < OWN; ++i) {
            double vc[AT];
#pragma unroll
            for (int a = 0; a < AT; ++a) vc[a] = Vcur[t[i][a]];
            double vmax = 0.0;
#pragma unroll
            for (int a = 0; a < AT; ++a) {
                qn[i][a] = r[i][a] + p.gamma * (term_s[i] ? 0.0 : vc[a]);
                nc |= own[i] && !isclose_np(qp[i][a], qn[i][a], p.rtol, p.atol);
                if (a == 0 || qn[i][a] > vmax) vmax = qn[i][a];
            }
            if (own[i]) Vnext[tid + i * BLOCK] = vmax;
        }
        // every thread has read V_k and written its part of V_{k+1}; the vote decides the sweep for the whole MDP
        // (a single-wave workgroup votes by ballot: the hardware issues one wave's LDS accesses in order, so no s_barrier is
        // needed -- but the COMPILER must not move the V_{k+1} stores below the next sweep's gathers of it: a workgroup-scope
        // fence + wave barrier pins the order at no run-time cost beyond an s_waitcnt lgkmcnt(0))
        if (BLOCK == 64) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const bool moved = BLOCK == 64 ? __any(nc ? 1 : 0) != 0 : __syncthreads_or(nc ? 1 : 0) != 0;
        if (!moved) { sweeps = k + 1; break; } // allclose(Q_k, Q_{k+1}): return Q_k = qp
#pragma unroll
        for (int i = 0; i < OWN; ++i)
#pragma unroll
            for (int a = 0; a < AT; ++a) qp[i][a] = qn[i][a];
    }
    if (tid == 0 && p.sweeps_out) p.sweeps_out[b] = sweeps;
    if (p.Q_out)
#pragma unroll
        for (int i = 0; i < OWN; ++i)
            if (own[i])
#pragma unroll
                for (int a = 0; a < AT; ++a) p.Q_out[(base + tid + i * BLOCK) * AT + a] = qp[i][a];
}

// CLUSTER form (round 6): K WORKGROUPS PER MDP, for batches of FEW LARGE MDPs that would leave most of the chip idle with one
// workgroup each (64 x S = 10 000: 64 of 256 CUs).  It is the register form -- a thread keeps the rows and the last Q row of its
// OWN states (those of its workgroup's slice of the MDP) for the whole solve, V double-buffered in LDS, all of it in every
// workgroup -- plus an exchange per sweep: a workgroup publishes its slice of V_{k+1} with write-through (sc1) stores, the
// cluster meets at a counter (one word per MDP and sweep: arrivals in the low half, "something moved" votes in the high half
// -- the barrier IS np.allclose's verdict, so every workgroup takes the same exit), and each workgroup fills the rest of its LDS
// copy with L1-bypassing (sc1) loads (cdna_hip_programming.md Guideline 16: sc1 stores -> s_waitcnt vmcnt(0) -> barrier ->
// relaxed agent fetch_add; relaxed polls; sc1 reads).  A cluster's workgroups are block ids 8 apart: the dispatcher places block
// b on XCD b % 8, so they share an L2 (a speed choice only; any placement is correct).  They must be co-resident: the host
// launches at most one workgroup per CU, and a spin limit turns a cluster that never met into sweeps_out = -1, which the
// follow-up launch (the global-memory workgroup form with only_failed) solves again.
template <int AT, int OWN>
__global__ __launch_bounds__(1024) void vi_det_batch_cluster(ViBatchArgs p, int K, double *__restrict__ Vx, unsigned *__restrict__ words_all,
                                                             unsigned need, unsigned spin_limit)
{
    extern __shared__ __attribute__((aligned(16))) double lds_v[];
    constexpr int NT = 1024;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int b = (slot / K) * 8 + xcd, part = slot % K, tid = threadIdx.x, S = p.Sb;
    if (b >= p.N) return;
    const int Sp = (S + K - 1) / K, s_lo = part * Sp, s_hi = min(S, s_lo + Sp), n_mine = max(s_hi - s_lo, 0);
    const long base = (long)b * S;
    double *V0 = lds_v, *V1 = lds_v + S;
    unsigned *bc = reinterpret_cast<unsigned *>(lds_v + 2 * (long)S);   // [4]: the verdict, double-buffered
    double *Vxb = Vx + (long)b * 2 * S;                                 // [2][S]: V_k published in slot k & 1
    gu32_t *words = (gu32_t *)(words_all + (long)b * p.iterations);
    int32_t t[OWN][AT];
    double r[OWN][AT], qp[OWN][AT];
    bool term_s[OWN], own[OWN];
#pragma unroll
    for (int i = 0; i < OWN; ++i) {
        const int s = s_lo + tid + i * NT;
        own[i] = s < s_hi;
        const long sa0 = (base + (own[i] ? s : 0)) * AT;
        term_s[i] = (p.term && own[i]) ? p.term[base + s] != 0 : false;
#pragma unroll
        for (int a = 0; a < AT; ++a) {
            t[i][a] = p.T[sa0 + a] - (int32_t)base;
            r[i][a] = p.R[sa0 + a];
            qp[i][a] = 0.0;                                  // Q_0 = 0 (value_iteration.py:43)
        }
    }
    for (int i = tid; i < S; i += NT) V0[i] = 0.0;
    __syncthreads();
    int sweeps = p.iterations;
    for (int k = 0; k < p.iterations; ++k) {
        const double *Vcur = (k & 1) ? V1 : V0;
        double *Vnext = (k & 1) ? V0 : V1;
        double *Vpub = Vxb + (long)((k + 1) & 1) * S;
        bool nc = false;
        double qn[OWN][AT];
#pragma unroll
        for (int i = 0; i < OWN; ++i) {
            double vc[AT];
#pragma unroll
            for (int a = 0; a < AT; ++a) vc[a] = Vcur[t[i][a]];
            double vmax = 0.0;
#pragma unroll
            for (int a = 0; a < AT; ++a) {
                qn[i][a] = r[i][a] + p.gamma * (term_s[i] ? 0.0 : vc[a]);
                nc |= own[i] && !isclose_np(qp[i][a], qn[i][a], p.rtol, p.atol);
                if (a == 0 || qn[i][a] > vmax) vmax = qn[i][a];
            }
            if (own[i]) {
                const int s = s_lo + tid + i * NT;
                Vnext[s] = vmax;
                __hip_atomic_store((gu64_t *)(Vpub + s), (unsigned long long)__double_as_longlong(vmax), MP_RLX_AGENT); // (8 bytes: an sc1 store)
            }
        }
        // ---- the cluster's barrier and the sweep's verdict in one word
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int agg = __syncthreads_or(nc ? 1 : 0);
        unsigned *bck = bc + 2 * (k & 1);
        if (tid == 0) {
            __hip_atomic_fetch_add(words + k, 1u + (agg ? 0x10000u : 0u), MP_RLX_AGENT);
            unsigned spins = 0, w, bad = 0;
            while (((w = __hip_atomic_load(words + k, MP_RLX_AGENT)) & 0xffffu) < need) {   // (need = K)
                __builtin_amdgcn_s_sleep(1);
                if (++spins > spin_limit) { bad = 1; break; }
            }
            bck[0] = bad; bck[1] = w >> 16;
        }
        __syncthreads();
        if (bck[0]) { sweeps = -1; break; }                  // (the cluster never met: not co-resident)
        if (bck[1] == 0) { sweeps = k + 1; break; }          // allclose(Q_k, Q_{k+1}) over the whole MDP: return Q_k = qp
#pragma unroll
        for (int i = 0; i < OWN; ++i)
#pragma unroll
            for (int a = 0; a < AT; ++a) qp[i][a] = qn[i][a];
        // ---- the other workgroups' slices of V_{k+1}, past the L1
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)Vpub, 0, S * 8, 0x00020000);
        for (int i = tid; i < S - n_mine; i += NT) {
            const int s = i < s_lo ? i : i + n_mine;
            const uint2 w = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, s * 8, 0, 16));
            Vnext[s] = __hiloint2double((int)w.y, (int)w.x);
        }
        __syncthreads();
    }
    if (tid == 0 && part == 0 && p.sweeps_out) p.sweeps_out[b] = sweeps;
    if (p.Q_out && sweeps >= 0)
#pragma unroll
        for (int i = 0; i < OWN; ++i)
            if (own[i])
#pragma unroll
                for (int a = 0; a < AT; ++a) p.Q_out[(base + s_lo + tid + i * NT) * AT + a] = qp[i][a];
}

// WORKGROUP form (any |A|, Sb beyond the register form): 1024 threads walk the states of their MDP; the tables stream from
// global memory (L2 / MALL resident across sweeps: 12 B per (s, a)); Q_k for the allclose test is recomputed from V_{k-1}, as
// the chained single-MDP sweeps do.  VLDS: V_{k-1} and V_k live in LDS and a thread holds its V_{k+1} values in registers
// until every gather of the sweep is done (two barriers per sweep, 16 * Sb bytes of LDS: Sb <= 10 200); otherwise three
// buffers per MDP in global memory (a workgroup's own writes are visible to it after a barrier).
constexpr int kViBatchOwn = 10; // states per thread the VLDS form holds V_{k+1} for: Sb <= 10 * 1024
template <int AT, bool VLDS>
__global__ __launch_bounds__(1024) void vi_det_batch_wg(ViBatchArgs p)
{
    extern __shared__ __attribute__((aligned(16))) double lds_v[];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, S = p.Sb, A = AT > 0 ? AT : p.A;
    if (p.only_failed && p.sweeps_out[b] != -1) return;      // (uniform: before any barrier)
    const long base = (long)b * S;
    double *Vb = VLDS ? lds_v : p.Vglobal + (long)b * 3 * S; // VLDS: [2][S]; global: [3][S]
    constexpr int NB = VLDS ? 2 : 3;
    for (int i = tid; i < NB * S; i += nt) Vb[i] = 0.0;
    __syncthreads();
    constexpr int AR = AT > 0 ? AT : 1;
    auto qrow = [&](const double *V, int s, bool term_s, double *out) {
        const long sa0 = (base + s) * A;
        if (AT > 0) {
            int32_t tt[AR];
            double rr[AR];
#pragma unroll
            for (int a = 0; a < AR; ++a) { tt[a] = p.T[sa0 + a] - (int32_t)base; rr[a] = p.R[sa0 + a]; }
#pragma unroll
            for (int a = 0; a < AR; ++a) out[a] = rr[a] + p.gamma * (term_s ? 0.0 : V[tt[a]]);
        } else {
            for (int a = 0; a < A; ++a) out[a] = p.R[sa0 + a] + p.gamma * (term_s ? 0.0 : V[p.T[sa0 + a] - (int32_t)base]);
        }
    };
    int j = p.iterations, sweeps = p.iterations;
    for (int k = 0; k < p.iterations; ++k) {
        // VLDS: V_k in buffer k & 1, V_{k-1} in the other, V_{k+1} replaces V_{k-1} after the gathers; global: ring of three
        const double *Vcur = Vb + (long)(VLDS ? (k & 1) : (k % 3)) * S;
        const double *Vprev = Vb + (long)(VLDS ? ((k + 1) & 1) : ((k + 2) % 3)) * S;
        double *Vnext = Vb + (long)(VLDS ? ((k + 1) & 1) : ((k + 1) % 3)) * S;
        bool nc = false;
        double vown[kViBatchOwn];
        int o = 0;
        for (int s = tid; s < S; s += nt, ++o) {
            const bool term_s = p.term ? p.term[base + s] != 0 : false;
            double vmax = 0.0;
            if (AT > 0) {
                double qn[AR], qo[AR];
                qrow(Vcur, s, term_s, qn);
                if (k > 0) qrow(Vprev, s, term_s, qo);
                vmax = qn[0];
#pragma unroll
                for (int a = 0; a < AR; ++a)
                    if (a < A) {
                        nc |= !isclose_np(k == 0 ? 0.0 : qo[a], qn[a], p.rtol, p.atol);
                        if (a > 0 && qn[a] > vmax) vmax = qn[a];
                    }
            } else { // any number of actions: one (s, a) at a time
                const long sa0 = (base + s) * A;
                for (int a = 0; a < A; ++a) {
                    const int tn = p.T[sa0 + a] - (int32_t)base;
                    const double r = p.R[sa0 + a];
                    const double qn = r + p.gamma * (term_s ? 0.0 : Vcur[tn]);
                    const double qo = k == 0 ? 0.0 : r + p.gamma * (term_s ? 0.0 : Vprev[tn]);
                    nc |= !isclose_np(qo, qn, p.rtol, p.atol);
                    if (a == 0 || qn > vmax) vmax = qn;
                }
            }
            if (VLDS) {
#pragma unroll
                for (int i = 0; i < kViBatchOwn; ++i) vown[i] = o == i ? vmax : vown[i];
            } else {
                Vnext[s] = vmax;
            }
        }
        if (__syncthreads_or(nc ? 1 : 0) == 0) { j = k; sweeps = k + 1; break; } // (V_{k-1} is still in place: Q_k below)
        if (VLDS) {
            o = 0;
            for (int s = tid; s < S; s += nt, ++o) {
                double v = vown[0];
#pragma unroll
                for (int i = 1; i < kViBatchOwn; ++i) v = o == i ? vown[i] : v;
                Vnext[s] = v;
            }
            __syncthreads();
        }
    }
    if (tid == 0 && p.sweeps_out) p.sweeps_out[b] = sweeps;
    if (p.Q_out) {
        // the returned iterate Q_j = Bellman(V_{j-1}) (0 for j = 0): V_{j-1} is where sweep j - 1 read V_cur from
        const double *Vjm1 = Vb + (long)(VLDS ? ((j + 1) & 1) : ((j + 2) % 3)) * S;
        for (int s = tid; s < S; s += nt) {
            const bool term_s = p.term ? p.term[base + s] != 0 : false;
            if (AT > 0) {
                double qj[AR];
                if (j > 0) qrow(Vjm1, s, term_s, qj);
                for (int a = 0; a < A; ++a) p.Q_out[(base + s) * A + a] = j == 0 ? 0.0 : qj[a];
            } else {
                const long sa0 = (base + s) * A;
                for (int a = 0; a < A; ++a)
                    p.Q_out[sa0 + a] = j == 0 ? 0.0 : p.R[sa0 + a] + p.gamma * (term_s ? 0.0 : Vjm1[p.T[sa0 + a] - (int32_t)base]);
            }
        }
    }
}

// LANE-MAJOR copies of a batch model's tables for the streaming form below: Tt uint16 [N][A][Sb] (LOCAL next state), Rt double
// [N][A][Sb] -- thread t of an MDP's workgroup owns states t, t + 1024, ...: with the state index innermost every load of a
// wavefront is one contiguous 128- / 512-byte run (state-major rows of |A| = 5 doubles are 40 bytes apart per lane: every
// load instruction would touch twenty cache lines for one line's worth of data).
__global__ __launch_bounds__(256) void vi_batch_transpose(int N, int Sb, int A, const int32_t *__restrict__ T, const double *__restrict__ R,
                                                          const uint8_t *__restrict__ term, uint16_t *__restrict__ Tt,
                                                          double *__restrict__ Rt)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; // over [N][A][Sb]
    if (i >= (long)N * A * Sb) return;
    const int s = (int)(i % Sb);
    const long ba = i / Sb;
    const int a = (int)(ba % A), b = (int)(ba / A);
    const long src = ((long)b * Sb + s) * A + a;
    // bit 15: terminal[s] of the SOURCE state (Sb <= 10 240 here) -- the flag arrives with the row it masks instead of
    // costing the sweep a dependent byte load per state
    Tt[i] = (uint16_t)((T[src] - b * Sb) | (term && term[(long)b * Sb + s] ? 0x8000 : 0));
    Rt[i] = R[src];
}

// STREAMING workgroup form (|A| at compile time, 4096 < Sb and 16 * Sb bytes of LDS: Sb <= 10 200 -- the C2 shape, S = 10 000).
// Thread t owns states t, t + 1024, ...; a sweep streams the lane-major tables once (10 B per (s, a): every wave-load one
// contiguous run), the rows of the NEXT state requested before the gathers of the current one, so that the only
// dependent chain of a state is its LDS gathers.  V_k and V_{k+1} are double-buffered in LDS -- V_{k+1} is written where it
// belongs as soon as it is known, one barrier per sweep -- and every V is also written to a ring of three in global memory
// (L2-resident, 8 B per state and sweep): V_{k-1}, which only the allclose test and the returned iterate need, is read
// from there.  allclose(Q_k, Q_{k+1}) is an OR over all (s, a): the sweep first tests only the pairs of each thread's FIRST
// state (exactly, Q_k recomputed from V_{k-1}); any pair that moved settles the sweep for the whole MDP -- every sweep but
// the last few -- and only when none of them moved does a second pass test the rest.  Same decisions, same returned
// iterate as the chained sweeps, half the gathers.
template <int AT>
__global__ __launch_bounds__(1024) void vi_det_batch_wgr(ViBatchArgs p, const uint16_t *__restrict__ Tt, const double *__restrict__ Rt)
{
    extern __shared__ __attribute__((aligned(16))) double lds_v[];
    constexpr int NT = 1024;
    const int b = blockIdx.x, tid = threadIdx.x, S = p.Sb;
    const long base = (long)b * S;
    const int n_own = (S + NT - 1) / NT;
    // tables through buffer resources: a 32-bit per-lane offset (the state) and a scalar offset (the action's row) per load
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void *)(Rt + (long)b * AT * S), 0, AT * S * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t tres = __builtin_amdgcn_make_buffer_rsrc((void *)(Tt + (long)b * AT * S), 0, AT * S * 2, 0x00020000);
    auto rload = [&](int a, int s) -> double {
        const uint2 w = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rres, s * 8, a * S * 8, 0));
        return __hiloint2double((int)w.y, (int)w.x);
    };
    auto tload = [&](int a, int s) -> int { return (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(tres, s * 2, a * S * 2, 0); };
    double *Vb = lds_v;                                  // [2][S]: V_k in buffer k & 1
    double *Vg = p.Vglobal + (long)b * 3 * S;            // [3][S]: V_k in slot k % 3
    __shared__ int vote[3];
    for (int i = tid; i < 2 * S; i += NT) Vb[i] = 0.0;
    for (int i = tid; i < 3 * S; i += NT) Vg[i] = 0.0;
    if (tid < 3) vote[tid] = 0;
    __syncthreads();
    int j = p.iterations, sweeps = p.iterations;
    for (int k = 0; k < p.iterations; ++k) {
        const double *Vcur = Vb + (long)(k & 1) * S;
        double *Vnext = Vb + (long)((k + 1) & 1) * S;
        const double *Vprev = Vg + (long)((k + 2) % 3) * S;
        double *Vnext_g = Vg + (long)((k + 1) % 3) * S;
        bool nc = false;
        // rows requested ONE state ahead (two register stages): the only dependent chain of a state is its LDS gathers.
        // (Three stages ahead measured SLOWER -- 1.41 against 1.25 ms for 64 MDPs x 101 sweeps: the sweep is not waiting for
        // the tables, it is bound by issue + LDS gathers + the barrier.)
        int tn[AT];
        double rn[AT];
        auto request = [&](int s) {
            const int sc = s < S ? s : (tid < S ? tid : 0);     // (beyond the last state: re-read a row that is there)
#pragma unroll
            for (int a = 0; a < AT; ++a) { tn[a] = tload(a, sc); rn[a] = rload(a, sc); }
        };
        request(tid);
        for (int i = 0; i < n_own; ++i) {
            const int s = tid + i * NT;
            int t[AT];
            double r[AT];
#pragma unroll
            for (int a = 0; a < AT; ++a) { t[a] = tn[a]; r[a] = rn[a]; }
            request(s + NT);
            const bool term_s = (t[0] & 0x8000) != 0;           // (rides in the row: vi_batch_transpose)
#pragma unroll
            for (int a = 0; a < AT; ++a) t[a] &= 0x7fff;
            double vc[AT], vp[AT];
#pragma unroll
            for (int a = 0; a < AT; ++a) vc[a] = Vcur[t[a]];
            if (i == 0 && k > 0) {
#pragma unroll
                for (int a = 0; a < AT; ++a) vp[a] = Vprev[t[a]];
            }
            double vmax = 0.0;
#pragma unroll
            for (int a = 0; a < AT; ++a) {
                const double qn = r[a] + p.gamma * (term_s ? 0.0 : vc[a]);
                if (i == 0) {
                    const double qo = k == 0 ? 0.0 : r[a] + p.gamma * (term_s ? 0.0 : vp[a]);
                    nc |= s < S && !isclose_np_sel(qo, qn, p.rtol, p.atol);
                }
                if (a == 0 || qn > vmax) vmax = qn;
            }
            if (s < S) { Vnext[s] = vmax; Vnext_g[s] = vmax; }
        }
        // the vote: ONE barrier (a wave that saw movement raises the sweep's flag; three flags rotate so that the reset of a
        // flag lies a whole sweep away from its last reader and its next writer)
        if (__any(nc ? 1 : 0) && (tid & 63) == 0) vote[k % 3] = 1;
        __syncthreads();
        bool moved = vote[k % 3] != 0;
        if (tid == 0) vote[(k + 2) % 3] = 0;
        if (!moved) { // none of the first states' pairs moved: test the others (the last few sweeps only)
            bool nc2 = false;
            for (int s = tid + NT; s < S; s += NT) {
#pragma unroll
                for (int a = 0; a < AT; ++a) {
                    const int tw = tload(a, s);
                    const bool term_s = (tw & 0x8000) != 0;
                    const int t = tw & 0x7fff;
                    const double r = rload(a, s);
                    const double qn = r + p.gamma * (term_s ? 0.0 : Vcur[t]);
                    const double qo = k == 0 ? 0.0 : r + p.gamma * (term_s ? 0.0 : Vprev[t]);
                    nc2 |= !isclose_np_sel(qo, qn, p.rtol, p.atol);
                }
            }
            moved = __syncthreads_or(nc2 ? 1 : 0) != 0;
        }
        if (!moved) { j = k; sweeps = k + 1; break; }
    }
    if (tid == 0 && p.sweeps_out) p.sweeps_out[b] = sweeps;
    if (p.Q_out) {
        const double *Vjm1 = Vg + (long)((j + 2) % 3) * S; // V_{j-1}
        for (int s = tid; s < S; s += NT) {
#pragma unroll
            for (int a = 0; a < AT; ++a) {
                const int tw = tload(a, s);
                p.Q_out[(base + s) * AT + a] = j == 0 ? 0.0 : rload(a, s) + p.gamma * ((tw & 0x8000) ? 0.0 : Vjm1[tw & 0x7fff]);
            }
        }
    }
}

template <int AT>
static int vi_batch_launch(mp_ctx *ctx, ViBatchArgs &q, hipStream_t st, const char **variant)
{
    const int S = q.Sb;
    const unsigned grid = (unsigned)q.N;
    if constexpr (AT > 0) {
        const size_t lds = (size_t)2 * S * sizeof(double);
#define MP_VB(own, block)                                                                                              \
    if (S <= (own) * (block)) {                                                                                        \
        hipLaunchKernelGGL((vi_det_batch_reg<AT, own, block>), dim3(grid), dim3(block), lds, st, q);                   \
        *variant = "vi_batch_reg<" #own "," #block ">";                                                                \
        return MP_OK;                                                                                                  \
    }
        if (!getenv("MP_VI_BATCH_NO_REG")) {
            MP_VB(1, 64) MP_VB(2, 64) MP_VB(2, 128) MP_VB(2, 256) MP_VB(4, 256) MP_VB(4, 512)
            if constexpr (AT <= 4) { MP_VB(4, 1024) }
        }
#undef MP_VB
    }
    if constexpr (AT > 0 && AT <= 6) {
        // CLUSTER form: the batch leaves CUs idle with one workgroup per MDP -> K = 2, 4 or 8 workgroups per MDP, at most one per CU
        // (each takes a CU's LDS: co-resident by construction on an otherwise idle device), each thread owning <= 3 states
        const long cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
        int K = 1;
        while (K < 8 && (long)q.N * (K * 2) <= cus) K *= 2;
        if (const char *e = getenv("MP_VI_BATCH_CLUSTER")) K = atoi(e);    // 0 / 1: off; 2, 4, 8: forced
        const int own = K > 1 ? ((S + K - 1) / K + 1023) / 1024 : 99;
        const size_t lds = (size_t)2 * S * sizeof(double) + 16;
        if (K > 1 && K <= 8 && own <= 3 && lds <= kLdsBytes && q.iterations > 0) {
            double *Vx = nullptr;
            unsigned *words = nullptr;
            MP_TRY(ws_get(ctx, WS_VI1, (size_t)q.N * 3 * S, &Vx));          // (the cluster uses [N][2][S]; the fallback [N][3][S])
            MP_TRY(ws_get(ctx, WS_VI3, (size_t)q.N * q.iterations, &words));
            MP_HIP(hipMemsetAsync(words, 0, (size_t)q.N * q.iterations * sizeof(unsigned), st));
            const unsigned cgrid = (unsigned)((q.N + 7) / 8) * 8 * K;
            // (test hook MP_VI_BATCH_CLUSTER_NEVER_MEETS=1: a cluster waits for one arrival too many and gives up after a short spin)
            const bool hook = getenv("MP_VI_BATCH_CLUSTER_NEVER_MEETS") != nullptr;
            const unsigned need = (unsigned)K + (hook ? 1u : 0u), spin = hook ? 2000u : kSpinLimit;
#define MP_VC(o)                                                                                                            \
    {                                                                                                                       \
        MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vi_det_batch_cluster<AT, o>),                             \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                 \
        hipLaunchKernelGGL((vi_det_batch_cluster<AT, o>), dim3(cgrid), dim3(1024), lds, st, q, K, Vx, words, need, spin);  \
    }
            if (own == 1) MP_VC(1) else if (own == 2) MP_VC(2) else MP_VC(3)
#undef MP_VC
            // clusters that never met (a busy device): solved again by the global-memory workgroup form (returns at once otherwise)
            ViBatchArgs f = q;
            f.only_failed = 1; f.Vglobal = Vx;
            hipLaunchKernelGGL((vi_det_batch_wg<AT, false>), dim3(grid), dim3(1024), 0, st, f);
            *variant = K == 2 ? "vi_batch_cluster2" : (K == 4 ? "vi_batch_cluster4" : "vi_batch_cluster8");
            return MP_OK;
        }
    }
    if constexpr (AT > 0) {
        if (S <= kViBatchOwn * 1024 && (size_t)2 * S * sizeof(double) <= kLdsBytes - 512 && !getenv("MP_VI_BATCH_NO_VLDS") &&
            !getenv("MP_VI_BATCH_NO_WGR")) {
            // lane-major copies of the tables (a few microseconds; rebuilt on every call: the tables of a batch change
            // between two steps of its episodes)
            const size_t npairs = (size_t)q.N * S * AT;
            uint16_t *Tt = nullptr;
            double *Rt = nullptr;
            MP_TRY(ws_get(ctx, WS_VI2, npairs, &Rt));
            MP_TRY(ws_get(ctx, WS_VI4, npairs, &Tt));
            MP_TRY(ws_get(ctx, WS_VI1, (size_t)q.N * 3 * S, &q.Vglobal));
            hipLaunchKernelGGL(vi_batch_transpose, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, q.N, S, AT, q.T, q.R, q.term, Tt, Rt);
            const size_t lds = (size_t)2 * S * sizeof(double);
            MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vi_det_batch_wgr<AT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds));
            hipLaunchKernelGGL((vi_det_batch_wgr<AT>), dim3(grid), dim3(1024), lds, st, q, (const uint16_t *)Tt, (const double *)Rt);
            *variant = "vi_batch_wg_stream";
            return MP_OK;
        }
    }
    if (S <= kViBatchOwn * 1024 && (size_t)2 * S * sizeof(double) <= kLdsBytes - 512 && !getenv("MP_VI_BATCH_NO_VLDS")) {
        const size_t lds = (size_t)2 * S * sizeof(double);
        if (lds > 64 * 1024)
            MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vi_det_batch_wg<AT, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds));
        hipLaunchKernelGGL((vi_det_batch_wg<AT, true>), dim3(grid), dim3(1024), lds, st, q);
        *variant = "vi_batch_wg_lds";
        return MP_OK;
    }
    MP_TRY(ws_get(ctx, WS_VI1, (size_t)q.N * 3 * S, &q.Vglobal));
    hipLaunchKernelGGL((vi_det_batch_wg<AT, false>), dim3(grid), dim3(1024), 0, st, q);
    *variant = "vi_batch_wg_global";
    return MP_OK;
}

static int vi_solve_batch(mp_ctx *ctx, mp_model *m, double gamma, int iterations, double rtol, double atol, double *Q_out,
                          int32_t *sweeps_out, int mem)
{
    if (!ctx || !m) return fail(MP_ERR_ARG, "mp_vi_solve_batch: NULL ctx/model");
    if (iterations < 0) return fail(MP_ERR_ARG, "mp_vi_solve_batch: iterations < 0");
    if (m->mode != MP_MODE_DETERMINISTIC || !m->T || m->M != 1)
        return fail(MP_ERR_MODE, "mp_vi_solve_batch: deterministic table models (mp_model_load_table_batch, or one MDP with M = 1) only");
    if (mem != MP_MEM_HOST && mem != MP_MEM_DEVICE) return fail(MP_ERR_ARG, "mp_vi_solve_batch: unknown mem flags %d", mem);
    MP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int N = m->NB, Sb = m->Sb > 0 ? m->Sb : m->S, A = m->A;
    const size_t nq = (size_t)N * Sb * A;
    double *dQ = nullptr;
    int32_t *dSw = nullptr;
    MP_TRY(stage_out_alloc(ctx, WS_IO0, Q_out, nq, mem, &dQ));
    MP_TRY(stage_out_alloc(ctx, WS_IO2, sweeps_out, (size_t)N, mem, &dSw));
    ViBatchArgs q;
    memset(&q, 0, sizeof(q));
    q.N = N; q.Sb = Sb; q.A = A; q.iterations = iterations; q.T = m->T; q.R = m->R; q.term = m->term;
    q.gamma = gamma; q.rtol = rtol; q.atol = atol; q.Q_out = dQ; q.sweeps_out = dSw;
    const char *variant = "";
    MP_TRY(kernels_begin(ctx));
    switch (A) {
    case 2: MP_TRY(vi_batch_launch<2>(ctx, q, st, &variant)); break;
    case 3: MP_TRY(vi_batch_launch<3>(ctx, q, st, &variant)); break;
    case 4: MP_TRY(vi_batch_launch<4>(ctx, q, st, &variant)); break;
    case 5: MP_TRY(vi_batch_launch<5>(ctx, q, st, &variant)); break;
    case 6: MP_TRY(vi_batch_launch<6>(ctx, q, st, &variant)); break;
    case 8: MP_TRY(vi_batch_launch<8>(ctx, q, st, &variant)); break;
    default: MP_TRY(vi_batch_launch<0>(ctx, q, st, &variant)); break;
    }
    MP_TRY(kernels_end(ctx, 1));
    snprintf(ctx->last_variant, sizeof(ctx->last_variant), "%s", variant);
    MP_HIP(hipGetLastError());
    MP_TRY(stage_out_copy(ctx, Q_out, dQ, nq, mem));
    MP_TRY(stage_out_copy(ctx, sweeps_out, dSw, (size_t)N, mem));
    if (mem == MP_MEM_HOST) MP_HIP(hipStreamSynchronize(st));
    return MP_OK;
}

// One Bellman backup Q = min_m (R_m + gamma * mask(T_m . V)) of a dense (possibly row-block) model.
static int vi_backup(mp_ctx *ctx, mp_model *m, double gamma, int robust, const double *V, double *Q, int mem)
{
    if (!ctx || !m || !V || !Q) return fail(MP_ERR_ARG, "mp_vi_backup: NULL argument");
    if (m->mode != MP_MODE_STOCHASTIC) return fail(MP_ERR_MODE, "mp_vi_backup: dense models only");
    MP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const long SA = (long)m->S * m->A;
    double *dV = nullptr, *dQ = nullptr;
    MP_TRY(stage_in(ctx, WS_IO0, V, (size_t)m->Sc, mem, &dV));
    MP_TRY(stage_out_alloc(ctx, WS_IO1, Q, (size_t)SA, mem, &dQ));
    int32_t *flag = nullptr;
    MP_TRY(ws_get(ctx, WS_VI3, 4, &flag));
    ViGenArgs a;
    memset(&a, 0, sizeof(a));
    a.M = robust ? m->M : 1; a.S = m->S; a.A = m->A; a.Sc = m->Sc; a.robust = robust; a.k = 0;
    a.P = m->P; a.R = m->R; a.term = m->term; a.gamma = gamma; a.Vcur = dV; a.Qnext = dQ; a.notclose = flag;
    MP_TRY(kernels_begin(ctx));
    int launches = 0;
    MP_TRY(vi_dense_launch(ctx, a, st, &launches));
    MP_TRY(kernels_end(ctx, launches));
    MP_HIP(hipGetLastError());
    MP_TRY(stage_out_copy(ctx, Q, dQ, (size_t)SA, mem));
    if (mem == MP_MEM_HOST) MP_HIP(hipStreamSynchronize(st));
    return MP_OK;
}

} // namespace mp

extern "C" {

int mp_vi_backup(mp_ctx *ctx, mp_model *model, double gamma, int32_t robust, const double *V, double *Q, int32_t mem)
{
    return mp::vi_backup(ctx, model, gamma, robust ? 1 : 0, V, Q, mem);
}

int mp_vi_solve(mp_ctx *ctx, mp_model *model, double gamma, int32_t iterations, double rtol, double atol,
                int32_t robust, double *Q_out, int32_t *sweeps_out, int32_t mem)
{
    return mp::vi_run(ctx, model, gamma, iterations, rtol, atol, robust ? 1 : 0, 0, Q_out, nullptr, sweeps_out, mem);
}

int mp_vi_solve_batch(mp_ctx *ctx, mp_model *model, double gamma, int32_t iterations, double rtol, double atol,
                      double *Q_out, int32_t *sweeps_out, int32_t mem)
{
    return mp::vi_solve_batch(ctx, model, gamma, iterations, rtol, atol, Q_out, sweeps_out, mem);
}

int mp_vi_solve_v(mp_ctx *ctx, mp_model *model, double gamma, int32_t iterations, double rtol, double atol,
                  double *V_out, int32_t mem)
{
    return mp::vi_run(ctx, model, gamma, iterations, rtol, atol, 0, 1, nullptr, V_out, nullptr, mem);
}

int mp_vi_solve_v_robust(mp_ctx *ctx, mp_model *model, double gamma, int32_t iterations, double rtol, double atol,
                         double *V_out, int32_t mem)
{
    return mp::vi_run(ctx, model, gamma, iterations, rtol, atol, 1, 1, nullptr, V_out, nullptr, mem);
}

int mp_vi_dense_mode(mp_ctx *ctx, int32_t mode)
{
    if (!ctx) return mp::fail(MP_ERR_ARG, "mp_vi_dense_mode: NULL ctx");
    if (mode != MP_VI_DENSE_MFMA && mode != MP_VI_DENSE_EXACT) return mp::fail(MP_ERR_ARG, "mp_vi_dense_mode: mode %d", (int)mode);
    ctx->vi_dense_exact = mode == MP_VI_DENSE_EXACT ? 1 : 0;
    return MP_OK;
}

int mp_vi_exact_plan(int32_t n, int32_t cap_leaves, int32_t *leaves, int32_t cap_nodes, int32_t *nodes, int32_t cap_heights,
                     int32_t *hoff, int32_t *counts)
{
    if (n < 1 || !counts) return mp::fail(MP_ERR_ARG, "mp_vi_exact_plan: n = %d", (int)n);
    std::vector<int> lv, nd, ho, pc;
    int nbm = 1;
    mp::vi_exact_plan_host(n, lv, nd, ho, pc, &nbm);
    counts[0] = (int32_t)lv.size() / 2; counts[1] = (int32_t)nd.size() / 2; counts[2] = (int32_t)ho.size() - 1; counts[3] = nbm;
    if (leaves && (size_t)cap_leaves * 2 >= lv.size()) memcpy(leaves, lv.data(), lv.size() * sizeof(int));
    if (nodes && (size_t)cap_nodes * 2 >= nd.size()) memcpy(nodes, nd.data(), nd.size() * sizeof(int));
    if (hoff && (size_t)cap_heights + 1 >= ho.size()) memcpy(hoff, ho.data(), ho.size() * sizeof(int));
    return MP_OK;
}

int mp_vi_sweeps(mp_ctx *ctx, mp_model *model, double gamma, int32_t sweeps, int32_t robust)
{
    // negative tolerances: |a - b| <= atol + rtol |b| is never true, so no sweep is skipped
    return mp::vi_run(ctx, model, gamma, sweeps, -1.0, -1.0, robust ? 1 : 0, 0, nullptr, nullptr, nullptr,
                      MP_MEM_DEVICE);
}

} // extern "C"
