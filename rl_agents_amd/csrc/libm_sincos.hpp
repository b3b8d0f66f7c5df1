// libm_sincos.hpp -- sin / cos of SMALL arguments exactly as the host's libm computes them, on the device.
//
// gymnasium's CartPole (restated in rl_agents_amd/envs/cartpole.py; the reference's only pinned test drives it,
// tests/agents/tree_search/test_mcts.py:7-19) evaluates math.sin / math.cos of the pole angle, i.e. the C library's
// double-precision sin / cos.  glibc's are not correctly rounded (0.55 ULP), so "the" value is whatever its algorithm
// yields: the device math library's sincos differs in the last bit now and then, which made CartPole the one planner
// path with a tolerance (rounds 1-4: >= 99.5 % of roots identical).  This header restates the algorithm of glibc's
// dbl-64 sin / cos for |x| < 0.855469 -- the only range a pole angle can reach (|theta| < 0.21 rad until termination)
// -- operation for operation:
//   |x| < 2^-26 (sin) / 2^-27 (cos)   x / 1.0
//   |x| < 0.126 (sin only)             the odd Taylor polynomial s1 .. s5
//   else                               x = xk + r with xk = round(|x| * 128) / 128 taken from big + |x|; sin / cos of xk from
//                                      the double-double table (sincos_table.inc), short polynomials in r, the
//                                      angle-addition formula with a correction term
// in the TWO forms the library ships on x86-64 and selects by CPU (ifunc): with fused multiply-adds where the compiler
// contracted them (`FMA = true`: the instruction sequence of glibc 2.35's FMA variant, decoded from the library; any CPU
// with FMA + AVX2, i.e. every host an MI355X sits in) and with every operation rounded (`FMA = false`: the SSE2 / AVX
// variants).  The host side of the library checks at model-load time which of the two reproduces THIS host's sin / cos on
// a sample of angles and tells the kernel (mp_libm_sincos_variant); if neither does (another libm), the device falls back to
// its own sincos and the parity claim for CartPole is the old tolerance (reported, never silent).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace mp {

#define MP_SINCOS_ENTRIES 440
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ const double kSincosTab[MP_SINCOS_ENTRIES] = {
#include "sincos_table.inc"
};
#else
static const double kSincosTab[MP_SINCOS_ENTRIES] = {
#include "sincos_table.inc"
};
#endif

struct LibmConst {
    static constexpr double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ecep-7, s3 = -0x1.a01a019db08b8p-13,
                            s4 = 0x1.71de27b9a7ed9p-19, s5 = -0x1.addffc2fcdf59p-26;
    static constexpr double sn3 = -0x1.5555555555515p-3, sn5 = 0x1.11110e829872fp-7;
    static constexpr double cs2 = 0x1.0p-1, cs4 = -0x1.5555555555535p-5, cs6 = 0x1.6c16bedd9e239p-10;
    static constexpr double big = 0x1.8p+45;
};

__host__ __device__ __forceinline__ double libm_fma(double a, double b, double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __fma_rn(a, b, c);
#else
    return fma(a, b, c);
#endif
}
__host__ __device__ __forceinline__ uint32_t libm_high_abs(double x)
{
    uint64_t b;
    memcpy(&b, &x, 8);
    return (uint32_t)(b >> 32) & 0x7fffffffu;
}
__host__ __device__ __forceinline__ int libm_low_word(double x)
{
    uint64_t b;
    memcpy(&b, &x, 8);
    return (int)(uint32_t)b;
}

// true when |x| is inside the restated range
__host__ __device__ __forceinline__ bool libm_small(double x) { return libm_high_abs(x) < 0x3feb6000u; }

template <bool FMA>
__host__ __device__ inline double libm_sin_small(double x, const double *tab = kSincosTab)
{
    typedef LibmConst K;
    const uint32_t k = libm_high_abs(x);
    if (k < 0x3e500000u) return x;
    const double ax = fabs(x);
    if (ax < 0.126) { // TAYLOR_SIN (x * x, x, 0)
        const double xx = x * x;
        double poly, t;
        if (FMA) {
            poly = libm_fma(xx, K::s5, K::s4);
            poly = libm_fma(xx, poly, K::s3);
            poly = libm_fma(xx, poly, K::s2);
            poly = libm_fma(xx, poly, K::s1);
            t = libm_fma(x, poly, -0.0);
            t = libm_fma(t, xx, 0.0);
        } else {
            poly = ((((K::s5 * xx + K::s4) * xx + K::s3) * xx + K::s2) * xx) + K::s1;
            t = (poly * x - 0.5 * 0.0) * xx + 0.0;
        }
        return x + t;
    }
    const double dx = x <= 0 ? -0.0 : 0.0;
    const double u = K::big + ax;
    const double r = ax - (u - K::big);
    const int i4 = libm_low_word(u) << 2;
    const double sn = tab[i4], ssn = tab[i4 + 1], cs = tab[i4 + 2], ccs = tab[i4 + 3];
    const double xx = r * r;
    double s, c, cor;
    if (FMA) {
        const double p = libm_fma(xx, K::sn5, K::sn3);
        s = r + libm_fma(r * xx, p, dx);
        double q = libm_fma(xx, K::cs6, K::cs4);
        q = libm_fma(xx, q, K::cs2);
        c = libm_fma(r, dx, xx * q);
        const double e1 = libm_fma(s, ccs, ssn);
        const double e2 = libm_fma(-c, sn, e1);
        cor = libm_fma(s, cs, e2);
    } else {
        s = r + (dx + r * xx * (K::sn3 + xx * K::sn5));
        c = r * dx + xx * (K::cs2 + xx * (K::cs4 + xx * K::cs6));
        cor = (ssn + s * ccs - sn * c) + cs * s;
    }
    return copysign(sn + cor, x);
}

template <bool FMA>
__host__ __device__ inline double libm_cos_small(double x, const double *tab = kSincosTab)
{
    typedef LibmConst K;
    const uint32_t k = libm_high_abs(x);
    if (k < 0x3e400000u) return 1.0;
    const double ax = fabs(x);
    const double dx = x < 0 ? -0.0 : 0.0;
    const double u = K::big + ax;
    const double r = ax - (u - K::big) + dx;
    const int i4 = libm_low_word(u) << 2;
    const double sn = tab[i4], ssn = tab[i4 + 1], cs = tab[i4 + 2], ccs = tab[i4 + 3];
    const double xx = r * r;
    double s, c, cor;
    if (FMA) {
        const double p = libm_fma(xx, K::sn5, K::sn3);
        s = libm_fma(r * xx, p, r);
        double q = libm_fma(xx, K::cs6, K::cs4);
        q = libm_fma(xx, q, K::cs2);
        c = xx * q;
        const double e1 = libm_fma(-s, ssn, ccs);
        const double e2 = libm_fma(-c, cs, e1);
        cor = libm_fma(-s, sn, e2);
    } else {
        s = r + r * xx * (K::sn3 + xx * K::sn5);
        c = xx * (K::cs2 + xx * (K::cs4 + xx * K::cs6));
        cor = (ccs - s * ssn - cs * c) - sn * s;
    }
    return cs + cor;
}

// c ? a : b as two v_cndmask_b32 on the lane mask of c: an opaque operation, so both operands are computed in the same basic
// block (a plain ?: lets the compiler branch around the costlier operand: two more blocks, and the instruction scheduler
// interleaves dependency chains only inside a block)
__host__ __device__ __forceinline__ double libm_select(bool c, double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long m = __builtin_amdgcn_ballot_w64(c);
    const int a_lo = __double2loint(a), a_hi = __double2hiint(a), b_lo = __double2loint(b), b_hi = __double2hiint(b);
    int lo, hi;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(b_lo), "v"(a_lo), "s"(m));
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(b_hi), "v"(a_hi), "s"(m));
    return __hiloint2double(hi, lo);
#else
    return c ? a : b;
#endif
}

// The same two functions in ONE basic block (round 6): every regime's value is computed and the regime selects -- operation for
// operation the arithmetic above, so the results are the same bits.  A branch would end the block, and the CartPole rollout
// overlaps the sin / cos of step t + 1 with the accelerations of step t by letting the instruction scheduler interleave two
// independent dependency chains, which it only does inside a block (uct.hip).  `tab` index clamped: a speculative call past the
// episode's end (the pole already out of range) must not read outside the table.
template <bool FMA>
__host__ __device__ __forceinline__ void libm_sincos_small_flat(double x, double *s_out, double *c_out, const double *tab = kSincosTab)
{
    typedef LibmConst K;
    const uint32_t k = libm_high_abs(x);
    const double ax = fabs(x);
    // sin, |x| < 0.126: the odd Taylor polynomial
    double sin_taylor;
    {
        const double xx = x * x;
        double poly, t;
        if (FMA) {
            poly = libm_fma(xx, K::s5, K::s4);
            poly = libm_fma(xx, poly, K::s3);
            poly = libm_fma(xx, poly, K::s2);
            poly = libm_fma(xx, poly, K::s1);
            t = libm_fma(x, poly, -0.0);
            t = libm_fma(t, xx, 0.0);
        } else {
            poly = ((((K::s5 * xx + K::s4) * xx + K::s3) * xx + K::s2) * xx) + K::s1;
            t = (poly * x - 0.5 * 0.0) * xx + 0.0;
        }
        sin_taylor = x + t;
    }
    // the table path, shared: x = xk + r
    const double u = K::big + ax;
    const double r0 = ax - (u - K::big);
    int i4 = libm_low_word(u) << 2;
    i4 = i4 < 0 ? 0 : (i4 > MP_SINCOS_ENTRIES - 4 ? MP_SINCOS_ENTRIES - 4 : i4);
    const double sn = tab[i4], ssn = tab[i4 + 1], cs = tab[i4 + 2], ccs = tab[i4 + 3];
    // What the two table paths share.  cos works on r = r0 + dx with dx = +-0: that IS r0, bit for bit (r0 = ax - (u - big) is never
    // -0, and r0 + (+-0) = r0 otherwise), so xx, r xx, the two polynomials and xx q are common; and sin's c = fma(r, dx, xx q) (plain
    // form: r dx + xx q) equals xx q, bit for bit: r dx = +-0 changes nothing unless xx q = 0, i.e. r = 0, where both give +0.
    double sin_table, cos_table;
    {
        const double r = r0;
        const double xx = r * r;
        const double dxs = x <= 0 ? -0.0 : 0.0;
        double s_sin, s_cos, c, cor_sin, cor_cos;
        if (FMA) {
            const double p = libm_fma(xx, K::sn5, K::sn3);
            const double rxx = r * xx;
            s_sin = r + libm_fma(rxx, p, dxs);
            s_cos = libm_fma(rxx, p, r);
            double q = libm_fma(xx, K::cs6, K::cs4);
            q = libm_fma(xx, q, K::cs2);
            c = xx * q;
            cor_sin = libm_fma(s_sin, cs, libm_fma(-c, sn, libm_fma(s_sin, ccs, ssn)));
            cor_cos = libm_fma(-s_cos, sn, libm_fma(-c, cs, libm_fma(-s_cos, ssn, ccs)));
        } else {
            const double rp = r * xx * (K::sn3 + xx * K::sn5);
            s_sin = r + (dxs + rp);
            s_cos = r + rp;
            c = xx * (K::cs2 + xx * (K::cs4 + xx * K::cs6));
            cor_sin = (ssn + s_sin * ccs - sn * c) + cs * s_sin;
            cor_cos = (ccs - s_cos * ssn - cs * c) - sn * s_cos;
        }
        sin_table = copysign(sn + cor_sin, x);
        cos_table = cs + cor_cos;
    }
    // (selects the compiler cannot turn back into branches around the regimes' arithmetic -- it did, with the plain ternaries)
    *s_out = libm_select(k < 0x3e500000u, x, libm_select(ax < 0.126, sin_taylor, sin_table));
    // (libm answers 1.0 below 2^-27 without computing; the table path's value there IS 1.0 -- grid point 0 has sn = ssn = ccs = 0,
    // cs = 1, so cos_table = 1 - xx q with xx q < 2^-55 -- and no select is needed; tests compare the tiny angles too)
    *c_out = cos_table;
}

enum { SINCOS_DEVICE = 0, SINCOS_LIBM_FMA = 1, SINCOS_LIBM_PLAIN = 2 };

// sin and cos of x in the form `variant` names; outside the restated range (never for a pole angle) the math library's
// (tab: the table in whatever memory the caller staged it -- the CartPole kernel keeps a copy in LDS: four lookups per step
// sit on the dependency chain of a lone wave)
__host__ __device__ __forceinline__ void libm_sincos(int variant, double x, double *s, double *c, const double *tab = kSincosTab)
{
    if (variant == SINCOS_LIBM_FMA && libm_small(x)) {
        *s = libm_sin_small<true>(x, tab);
        *c = libm_cos_small<true>(x, tab);
    } else if (variant == SINCOS_LIBM_PLAIN && libm_small(x)) {
        *s = libm_sin_small<false>(x, tab);
        *c = libm_cos_small<false>(x, tab);
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
        sincos(x, s, c);
#else
        *s = sin(x);
        *c = cos(x);
#endif
    }
}

} // namespace mp
