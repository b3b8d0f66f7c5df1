// wave.hpp -- cross-lane reductions on a 64-lane wavefront with DPP (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace mp {

// The lane mask of a predicate as the compare's own SGPR pair.  (HIP's __ballot / __any take an int: the predicate is first
// materialised with v_cndmask and compared with zero again -- two vector instructions per call in kernels that count them.)
__device__ __forceinline__ unsigned long long ballot64(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }
__device__ __forceinline__ bool any64(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0ull; }


// ---- 32-bit DPP reductions written as ONE instruction per step: `v_op_dpp v, v, v` computes op(dpp(v), v) in place and
// lanes without a valid DPP source keep their value (bound_ctrl off = the lane is disabled for the instruction).  The
// builtin route (update_dpp, then the operation) costs a register copy, the DPP move and the operation per step -- three
// issue slots where the hardware needs one; a planner wave at 8 waves per SIMD is bound by VALU issue.  `s_nop 1`: a DPP
// read of a VGPR written by the previous VALU instruction needs two wait states, and the hazard recogniser does not
// look inside inline assembly.  Call with all 64 lanes active.
#define MP_DPP_STEP(op, v, ctrl) asm("s_nop 1\n\t" op " %0, %0, %0 " ctrl : "+v"(v))
#define MP_DPP_REDUCE_ROW(op, v)                                  \
    MP_DPP_STEP(op, v, "row_shr:1 row_mask:0xf bank_mask:0xf");   \
    MP_DPP_STEP(op, v, "row_shr:2 row_mask:0xf bank_mask:0xf");   \
    MP_DPP_STEP(op, v, "row_shr:4 row_mask:0xf bank_mask:0xf");   \
    MP_DPP_STEP(op, v, "row_shr:8 row_mask:0xf bank_mask:0xf")
#define MP_DPP_REDUCE_WAVE(op, v)                                 \
    MP_DPP_REDUCE_ROW(op, v);                                     \
    MP_DPP_STEP(op, v, "row_bcast:15 row_mask:0xa bank_mask:0xf"); \
    MP_DPP_STEP(op, v, "row_bcast:31 row_mask:0xc bank_mask:0xf")

// One DPP reduction step on a double: lanes without a valid DPP source read their own value (old = own), so the
// step leaves them unchanged.  v_max_f64 is written out because the planners' bounds are never NaN (rewards are
// range-checked, the only special value is -inf) and `fmax` would add a canonicalisation of both inputs per step.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double max_step(double u)
{
    const int lo = __double2loint(u), hi = __double2hiint(u);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const double ou = __hiloint2double(ohi, olo);
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(u), "v"(ou));
    return r;
}

__device__ __forceinline__ double bcast_lane(double u, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(u), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(u), lane);
    return __hiloint2double(hi, lo);
}

// every lane returns the wave-wide maximum (row_shr 1/2/4/8, row_bcast 15/31: lane 63 holds the result)
__device__ __forceinline__ double wave_max(double u)
{
    u = max_step<0x111, 0xf>(u);
    u = max_step<0x112, 0xf>(u);
    u = max_step<0x114, 0xf>(u);
    u = max_step<0x118, 0xf>(u);
    u = max_step<0x142, 0xa>(u);
    u = max_step<0x143, 0xc>(u);
    return bcast_lane(u, 63);
}

// every lane returns the maximum over lanes 0..15 (the first DPP row)
__device__ __forceinline__ double row0_max(double u)
{
    u = max_step<0x111, 0xf>(u);
    u = max_step<0x112, 0xf>(u);
    u = max_step<0x114, 0xf>(u);
    u = max_step<0x118, 0xf>(u);
    return bcast_lane(u, 15);
}

__device__ __forceinline__ int wave_min(int v)
{
    MP_DPP_REDUCE_WAVE("v_min_i32_dpp", v);
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int row0_min(int v)
{
    MP_DPP_REDUCE_ROW("v_min_i32_dpp", v);
    return __builtin_amdgcn_readlane(v, 15);
}

// ---- the same for values known to be >= +0.0 or -inf (the planners' bounds when gamma is in [0, 1) and the terminal
// reward is not negative): a lane without a DPP source reads 0 instead of its own value (bound_ctrl), so a step needs no
// copy of the value for the "old" operand -- four issue slots per row step instead of seven, and an instruction is five
// cycles of a planner wave's chain.  max(u, 0) = u for every real value; the maximum of lanes that are ALL -inf comes
// out as 0, which the callers detect by the id reduction finding no lane (wave_argmax_nonneg).
template <int CTRL>
__device__ __forceinline__ double max_step_zero(double u)
{
    const int lo = __double2loint(u), hi = __double2hiint(u);
    const int olo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    const int ohi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    const double ou = __hiloint2double(ohi, olo);
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(u), "v"(ou));
    return r;
}

// maximum of non-negative 32-bit integers, same idea: the DPP combiner folds each step into one v_max_i32_dpp
template <int CTRL>
__device__ __forceinline__ int imax_step_zero(int v)
{
    const int o = __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
    return o > v ? o : v;
}

// (row_bcast with every row enabled: row r also receives lane 15 of row r - 1 where the classic masks skip it -- one
// more operand of an idempotent maximum; row 0 / rows 0-1 read 0)
__device__ __forceinline__ void wave_argmax_nonneg(double &u, int &id)
{
    double m = max_step_zero<0x111>(u);
    m = max_step_zero<0x112>(m);
    m = max_step_zero<0x114>(m);
    m = max_step_zero<0x118>(m);
    m = max_step_zero<0x142>(m);
    m = max_step_zero<0x143>(m);
    m = bcast_lane(m, 63);
    // lowest id among the lanes that hold the maximum = largest (INT_MAX - id); 0 = this lane does not hold it
    int key = u == m ? 0x7fffffff - id : 0;
    key = imax_step_zero<0x111>(key);
    key = imax_step_zero<0x112>(key);
    key = imax_step_zero<0x114>(key);
    key = imax_step_zero<0x118>(key);
    key = imax_step_zero<0x142>(key);
    key = imax_step_zero<0x143>(key);
    key = __builtin_amdgcn_readlane(key, 63);
    id = 0x7fffffff - key;
    u = key == 0 ? -INFINITY : m; // no lane holds the "maximum": every lane was -inf
}

// unsigned maximum, zero-fill (see imax_step_zero)
template <int CTRL>
__device__ __forceinline__ unsigned umax_step_zero(unsigned v)
{
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
    return o > v ? o : v;
}

// The key form of the argmax (argmax_keys below: three separable 32-bit reductions, the fewest instructions -- the choice of
// the kernels that run 8 waves per SIMD) for bounds >= +0.0 or -inf: the bit pattern of such a double IS an order-preserving
// (signed high word, unsigned low word) key -- -inf has a negative high word -- so no transform, and every reduction step
// is one zero-fill DPP instruction.
__device__ __forceinline__ void wave_argmax_keys_nonneg(double &u, int &id)
{
    const int hi = __double2hiint(u);
    const unsigned lo = (unsigned)__double2loint(u);
    int kh = hi < 0 ? 0 : hi; // (-inf: below every real high word or equal to that of +0.0 -- then the low words / ids decide)
    kh = imax_step_zero<0x111>(kh); kh = imax_step_zero<0x112>(kh); kh = imax_step_zero<0x114>(kh);
    kh = imax_step_zero<0x118>(kh); kh = imax_step_zero<0x142>(kh); kh = imax_step_zero<0x143>(kh);
    const int mh = __builtin_amdgcn_readlane(kh, 63);
    const bool c1 = hi == mh;
#ifndef MP_ARGMAX_NO_EARLY_EXIT
    // ONE lane holds the maximal high word (the usual case: 20 mantissa bits rarely tie between leaves): the pair is that
    // lane's, read with two readlanes instead of twelve more reduction steps.  Wave-uniform branch.
    {
        const unsigned long long b1 = ballot64(c1);
        if (__popcll(b1) == 1) {
            const int src = __ffsll((long long)b1) - 1;
            id = __builtin_amdgcn_readlane(id, src);
            u = __hiloint2double(mh, __builtin_amdgcn_readlane((int)lo, src));
            return;
        }
    }
#endif
    unsigned l1 = c1 ? lo : 0u;
    l1 = umax_step_zero<0x111>(l1); l1 = umax_step_zero<0x112>(l1); l1 = umax_step_zero<0x114>(l1);
    l1 = umax_step_zero<0x118>(l1); l1 = umax_step_zero<0x142>(l1); l1 = umax_step_zero<0x143>(l1);
    const unsigned ml = (unsigned)__builtin_amdgcn_readlane((int)l1, 63);
#ifndef MP_ARGMAX_NO_EARLY_EXIT
    {
        const unsigned long long b2 = ballot64(c1 && lo == ml);
        if (__popcll(b2) == 1) {
            id = __builtin_amdgcn_readlane(id, __ffsll((long long)b2) - 1);
            u = __hiloint2double(mh, (int)ml);
            return;
        }
    }
#endif
    int key = (c1 && lo == ml) ? 0x7fffffff - id : 0;
    key = imax_step_zero<0x111>(key); key = imax_step_zero<0x112>(key); key = imax_step_zero<0x114>(key);
    key = imax_step_zero<0x118>(key); key = imax_step_zero<0x142>(key); key = imax_step_zero<0x143>(key);
    key = __builtin_amdgcn_readlane(key, 63);
    id = 0x7fffffff - key;
    u = key == 0 ? -INFINITY : __hiloint2double(mh, (int)ml); // no lane holds the "maximum": every lane was -inf
}

// ---- cross-lane argmax on (U, id): maximal U first, lowest id among equal U; every lane returns the pair.
// U is mapped to an order-preserving (signed high word, unsigned low word) key -- flip the magnitude bits of negative
// values -- and the argmax is three separable 32-bit reductions: the maximal high word, the maximal low word among the
// lanes that hold it, the minimal id among the lanes that hold both.  18 single-instruction DPP steps where the
// lexicographic (U, id) reduction took ~170 instructions and the (v_max_f64, v_min_i32) pair on builtins ~60.
// U is never NaN here (rewards are range-checked; the only special value is -inf); `+ 0.0` folds a -0.0 into +0.0 so
// that equal values have equal keys.
template <bool ROW0>
__device__ __forceinline__ void argmax_keys(double &u, int &id)
{
    const double uc = u + 0.0;
    const int hi = __double2hiint(uc), sg = hi >> 31;
    int kh = hi ^ (sg & 0x7fffffff);
    const unsigned kl = (unsigned)(__double2loint(uc) ^ sg);
    const int kh_own = kh;
    if (ROW0) { MP_DPP_REDUCE_ROW("v_max_i32_dpp", kh); } else { MP_DPP_REDUCE_WAVE("v_max_i32_dpp", kh); }
    const int mh = __builtin_amdgcn_readlane(kh, ROW0 ? 15 : 63);
    const bool c1 = kh_own == mh;
    const int ms = mh >> 31;
#ifndef MP_ARGMAX_NO_EARLY_EXIT
    if (!ROW0) {          // one lane holds the maximal high word: see wave_argmax_keys_nonneg
        const unsigned long long b1 = ballot64(c1);
        if (__popcll(b1) == 1) {
            const int src = __ffsll((long long)b1) - 1;
            id = __builtin_amdgcn_readlane(id, src);
            u = __hiloint2double(mh ^ (ms & 0x7fffffff), __builtin_amdgcn_readlane((int)kl, src) ^ ms);
            return;
        }
    }
#endif
    unsigned l1 = c1 ? kl : 0u;
    if (ROW0) { MP_DPP_REDUCE_ROW("v_max_u32_dpp", l1); } else { MP_DPP_REDUCE_WAVE("v_max_u32_dpp", l1); }
    const unsigned ml = (unsigned)__builtin_amdgcn_readlane((int)l1, ROW0 ? 15 : 63);
    int i2 = (c1 && kl == ml) ? id : 0x7fffffff;
    if (ROW0) { MP_DPP_REDUCE_ROW("v_min_i32_dpp", i2); } else { MP_DPP_REDUCE_WAVE("v_min_i32_dpp", i2); }
    id = __builtin_amdgcn_readlane(i2, ROW0 ? 15 : 63);
    u = __hiloint2double(mh ^ (ms & 0x7fffffff), (int)(ml ^ (unsigned)ms));
}

// Three dependent reductions (18 steps) but the fewest instructions: the choice of the kernels that run 8 waves per SIMD
// and are bound by VALU issue (OPD high-occupancy variant: 8192 roots 3.83 -> 3.60 ms).
__device__ __forceinline__ void wave_argmax_keys(double &u, int &id) { argmax_keys<false>(u, id); }

// Two dependent reductions (12 steps; v_max_f64 has no DPP form, so a step of the maximum is two DPP moves and the
// operation): the choice of the latency-bound kernels (one wave per SIMD: OPD with the bounds in LDS 1.15 vs 1.20 ms).
__device__ __forceinline__ void wave_argmax(double &u, int &id)
{
    const double m = wave_max(u);
    id = wave_min(u == m ? id : 0x7fffffff);
    u = m;
}

// the same over lanes 0..15 only (|A| <= 16 children, one per lane)
__device__ __forceinline__ void row0_argmax(double &u, int &id)
{
    const double m = row0_max(u);
    id = row0_min(u == m ? id : 0x7fffffff);
    u = m;
}

} // namespace mp
