// pcg64.hpp -- numpy.random.Generator(PCG64) on the device, bit for bit.
//
// The reference planners draw all randomness from planner.np_random, a numpy Generator over
// PCG64 (tree_search/abstract.py:124-131 via gymnasium.utils.seeding.np_random).  They consume
// it in two ways only:
//   * Node.random_argmax -> np_random.choice(indices) (abstract.py:304-311): for k >= 2 ties
//     one bounded draw integers(0, k) = Lemire multiply-shift with rejection on numpy's buffered
//     32-bit stream; for k == 1 nothing is drawn.
//   * np_random.choice(actions, 1, p=p) (mcts.py:172): one random() double, inverse CDF.
// Keeping the generator state per root on the device and stepping it exactly like numpy makes
// UCT plans bit-identical to the reference at equal seeds (no "tolerance parity" needed).
// PCG64 = 128-bit LCG (multiplier 0x2360ED051FC65DA44385DF649FCCF645) + XSL-RR 128/64 output.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mp {

struct Pcg64 {
    uint64_t s_hi, s_lo, inc_hi, inc_lo;
    uint32_t has_uint32, uinteger;

    __device__ __forceinline__ void load(const uint64_t *p)
    {
        // a PCG64 increment is odd by construction (numpy: inc = (seq << 1) | 1); forcing the bit keeps a
        // malformed (e.g. all-zero) record from turning the rejection loop of below() into a hang
        s_hi = p[0]; s_lo = p[1]; inc_hi = p[2]; inc_lo = p[3] | 1ULL;
        has_uint32 = (uint32_t)p[4]; uinteger = (uint32_t)p[5];
    }
    __device__ __forceinline__ void store(uint64_t *p) const
    {
        p[0] = s_hi; p[1] = s_lo; p[2] = inc_hi; p[3] = inc_lo;
        p[4] = has_uint32; p[5] = uinteger;
    }
    // d = a * m + c as one v_mad_u64_u32 (32 x 32 + 64 -> 64); the instruction's carry-out lane mask goes to a dead SGPR pair.
    // The multiplier limb sits in an SGPR (one constant-bus operand); written as inline assembly because the compiler
    // otherwise expands the schoolbook product into twice as many instructions (zero-extending moves, multiplies by a
    // literal 0 used as 64-bit adds: 60 VALU instructions per step against the 27 of this form).
    __device__ __forceinline__ static uint64_t mad64(uint32_t a, uint32_t m, uint64_t c)
    {
        uint64_t d, cy;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "s"(m), "v"(c));
        return d;
    }
    // state <- state * 0x2360ED051FC65DA44385DF649FCCF645 + inc (mod 2^128), by columns of 32-bit limbs: column k sums the
    // products a_i * m_j with i + j = k on top of the carry of column k - 1, each product one multiply-add into a 64-bit
    // accumulator (14 instructions; checked against Python integers, tests/test_host_logic.py restates it on the host).
    __device__ __forceinline__ void advance()
    {
        mul_add<0x9FCCF645u, 0x4385DF64u, 0x1FC65DA4u, 0x2360ED05u>(inc_lo, inc_hi);
    }
    // JUMP AHEAD.  k steps of the generator are one affine map: state <- A^k state + inc G_k with G_k = 1 + A + ... + A^(k-1)
    // (mod 2^128).  For k = 4 both factors are constants of the generator: A^4 = 0xF4DD417327DB7A9B_D194DFBE42D45771,
    // G_4 = 0x610E11A14B07E063_817FA187ADEFBA1C (checked against stepping in tests/test_host_logic.py and by every rollout of
    // the four-lanes-per-root kernel, whose generator states are compared with numpy's).
    // inc_g4(): inc * G_4, the additive term of a 4-step jump (constant per generator: computed once per plan).
    __device__ __forceinline__ void inc_g4(uint64_t &lo, uint64_t &hi) const
    {
        Pcg64 t = *this;
        t.s_lo = inc_lo; t.s_hi = inc_hi;
        t.mul_add<0xADEFBA1Cu, 0x817FA187u, 0x4B07E063u, 0x610E11A1u>(0, 0);
        lo = t.s_lo; hi = t.s_hi;
    }
    __device__ __forceinline__ void advance4(uint64_t g4_lo, uint64_t g4_hi)
    {
        mul_add<0x42D45771u, 0xD194DFBEu, 0x27DB7A9Bu, 0xF4DD4173u>(g4_lo, g4_hi);
    }
    // the same for k = 16: A^16 = 0xB6A4239F3B315F84_F6EF6D3D288C03C1, G_16 = 0x6ED699DB168FB143_A9072151352439F0 (Python integers;
    // every CartPole rollout of sixteen-lane replicas compares the generator it leaves with numpy's)
    __device__ __forceinline__ void inc_g16(uint64_t &lo, uint64_t &hi) const
    {
        Pcg64 t = *this;
        t.s_lo = inc_lo; t.s_hi = inc_hi;
        t.mul_add<0x352439F0u, 0xA9072151u, 0x168FB143u, 0x6ED699DBu>(0, 0);
        lo = t.s_lo; hi = t.s_hi;
    }
    __device__ __forceinline__ void advance16(uint64_t g16_lo, uint64_t g16_hi)
    {
        mul_add<0x288C03C1u, 0xF6EF6D3Du, 0x3B315F84u, 0xB6A4239Fu>(g16_lo, g16_hi);
    }
    // state <- A^n state + inc G_n for a PER-LANE n: the limbs of A^n and G_n come from a table (VGPR multipliers)
    __device__ __forceinline__ static uint64_t mad64v(uint32_t a, uint32_t m, uint64_t c)
    {
        uint64_t d, cy;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "v"(m), "v"(c));
        return d;
    }
    __device__ __forceinline__ static void mul128v(uint64_t x_lo, uint64_t x_hi, const uint32_t (&m)[4], uint64_t &r_lo, uint64_t &r_hi)
    {
        const uint32_t a0 = (uint32_t)x_lo, a1 = (uint32_t)(x_lo >> 32), a2 = (uint32_t)x_hi, a3 = (uint32_t)(x_hi >> 32);
        const uint64_t A0 = mad64v(a0, m[0], 0);
        const uint64_t B = mad64v(a0, m[1], A0 >> 32);
        const uint64_t A1 = mad64v(a1, m[0], B);
        const uint32_t c1 = A1 < B ? 1u : 0u;
        const uint64_t C = mad64v(a0, m[2], (A1 >> 32) | ((uint64_t)c1 << 32));
        const uint64_t D = mad64v(a1, m[1], C);
        const uint64_t A2 = mad64v(a2, m[0], D);
        uint64_t Q = mad64v(a0, m[3], 0);
        Q = mad64v(a1, m[2], Q);
        Q = mad64v(a2, m[1], Q);
        Q = mad64v(a3, m[0], Q);
        const uint32_t r3 = (uint32_t)Q + (uint32_t)(A2 >> 32);
        r_lo = (uint64_t)(uint32_t)A0 | ((uint64_t)(uint32_t)A1 << 32);
        r_hi = (uint64_t)(uint32_t)A2 | ((uint64_t)r3 << 32);
    }
    __device__ __forceinline__ void jump(const uint32_t (&an)[4], const uint32_t (&gn)[4])
    {
        uint64_t p_lo, p_hi, q_lo, q_hi;
        mul128v(s_lo, s_hi, an, p_lo, p_hi);
        mul128v(inc_lo, inc_hi, gn, q_lo, q_hi);
        const uint64_t lo = p_lo + q_lo;
        s_hi = p_hi + q_hi + (lo < p_lo ? 1ULL : 0ULL);
        s_lo = lo;
    }
    // state <- state * (m3:m2:m1:m0) + (add_hi:add_lo)  (mod 2^128), multiplier limbs as literals / SGPRs
    template <uint32_t M0, uint32_t M1, uint32_t M2, uint32_t M3>
    __device__ __forceinline__ void mul_add(uint64_t add_lo, uint64_t add_hi)
    {
        const uint32_t m0 = M0, m1 = M1, m2 = M2, m3 = M3;
        const uint32_t a0 = (uint32_t)s_lo, a1 = (uint32_t)(s_lo >> 32), a2 = (uint32_t)s_hi, a3 = (uint32_t)(s_hi >> 32);
        const uint64_t A0 = mad64(a0, m0, 0);                        // column 0
        const uint64_t B = mad64(a0, m1, A0 >> 32);                  // column 1: a0 m1 + carry (no overflow)
        const uint64_t A1 = mad64(a1, m0, B);                        //           + a1 m0 (65 bits: the carry is c1)
        // (c1 as a compare the compiler sees, not the instruction's own carry-out: on gfx950 a VALU that reads an SGPR
        // another VALU wrote needs two wait states in between, which only the compiler's hazard pass inserts)
        const uint32_t c1 = A1 < B ? 1u : 0u;
        const uint64_t C = mad64(a0, m2, (A1 >> 32) | ((uint64_t)c1 << 32)); // column 2 (carries beyond bit 127 are dropped)
        const uint64_t D = mad64(a1, m1, C);
        const uint64_t A2 = mad64(a2, m0, D);
        uint64_t Q = mad64(a0, m3, 0);                               // column 3: the low 32 bits only
        Q = mad64(a1, m2, Q);
        Q = mad64(a2, m1, Q);
        Q = mad64(a3, m0, Q);
        const uint32_t r3 = (uint32_t)Q + (uint32_t)(A2 >> 32);
        const uint64_t lo = (uint64_t)(uint32_t)A0 | ((uint64_t)(uint32_t)A1 << 32);
        uint64_t hi = (uint64_t)(uint32_t)A2 | ((uint64_t)r3 << 32);
        const uint64_t lo2 = lo + add_lo;
        hi += add_hi + (lo2 < lo ? 1ULL : 0ULL);
        s_hi = hi; s_lo = lo2;
    }
    __device__ __forceinline__ uint64_t next64()
    {
        advance();
        return output();
    }
    // the 64-bit output of the CURRENT state (numpy steps first, then outputs: next64 = advance + output)
    __device__ __forceinline__ uint64_t output() const
    {
        // XSL-RR: rotr64(hi ^ lo, hi >> 58) as two 32-bit funnel shifts (v_alignbit_b32 takes the shift modulo 32; a
        // rotation by 32 or more swaps the halves first)
        const uint32_t xl = (uint32_t)s_lo ^ (uint32_t)s_hi, xh = (uint32_t)(s_lo >> 32) ^ (uint32_t)(s_hi >> 32);
        const uint32_t rot = (uint32_t)(s_hi >> 58);
        const bool sw = rot >= 32u;
        const uint32_t a = sw ? xh : xl, b = sw ? xl : xh;           // rotating {b, a} by rot & 31
        const uint32_t lo = __builtin_amdgcn_alignbit(b, a, rot), hi = __builtin_amdgcn_alignbit(a, b, rot);
        return ((uint64_t)hi << 32) | lo;
    }
    __device__ __forceinline__ uint32_t next32()
    {
        if (has_uint32) {
            has_uint32 = 0;
            return uinteger;
        }
        const uint64_t n = next64();
        has_uint32 = 1;
        uinteger = (uint32_t)(n >> 32);
        return (uint32_t)n;
    }
    __device__ __forceinline__ double next_double()
    {
        return (double)(next64() >> 11) * (1.0 / 9007199254740992.0);
    }
    // Generator.integers(0, k) / Generator.choice(arange(k)), 1 <= k < 2^32
    __device__ __forceinline__ uint32_t below(uint32_t k)
    {
        if (k <= 1) return 0;
        uint64_t m = (uint64_t)next32() * k;
        uint32_t leftover = (uint32_t)m;
        if (leftover < k) {
            const uint32_t threshold = (uint32_t)(0u - k) % k;
            while (leftover < threshold) {
                m = (uint64_t)next32() * k;
                leftover = (uint32_t)m;
            }
        }
        return (uint32_t)(m >> 32);
    }
};

// The same generator for a WAVE-UNIFORM stream (uct_lone_kernel: one root per wavefront, every lane would hold the same state).
// Plain 64-bit C arithmetic on values made uniform by readfirstlane: the compiler keeps state and products in scalar registers
// (s_mul_i32 / s_mul_hi_u32 / s_addc_u32), which a lone wave issues several times faster than the quarter-rate v_mad_u64_u32
// chain of Pcg64::advance -- a generator step sits on the chain of every tie-break of a descent.
struct Pcg64U {
    uint64_t s_hi, s_lo, inc_hi, inc_lo;
    uint32_t has_uint32, uinteger;

    __device__ __forceinline__ static uint64_t uni(uint64_t x)
    {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
    }
    __device__ __forceinline__ void load(const uint64_t *p)
    {
        s_hi = uni(p[0]); s_lo = uni(p[1]); inc_hi = uni(p[2]); inc_lo = uni(p[3]) | 1ULL; // (odd increment: see Pcg64::load)
        has_uint32 = (uint32_t)uni(p[4]); uinteger = (uint32_t)uni(p[5]);
    }
    __device__ __forceinline__ void store(uint64_t *p) const
    {
        p[0] = s_hi; p[1] = s_lo; p[2] = inc_hi; p[3] = inc_lo;
        p[4] = has_uint32; p[5] = uinteger;
    }
    __device__ __forceinline__ static uint64_t mulhi(uint64_t a, uint64_t b)
    {
        const uint64_t a0 = (uint32_t)a, a1 = a >> 32, b0 = (uint32_t)b, b1 = b >> 32;
        const uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
        const uint64_t mid = (p00 >> 32) + (uint32_t)p01 + (uint32_t)p10;
        return p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
    }
    __device__ __forceinline__ void advance()
    {
        const uint64_t m_lo = 0x4385DF649FCCF645ULL, m_hi = 0x2360ED051FC65DA4ULL;
        const uint64_t lo = s_lo * m_lo;
        const uint64_t hi = mulhi(s_lo, m_lo) + s_lo * m_hi + s_hi * m_lo;
        const uint64_t lo2 = lo + inc_lo;
        // (the carry as the majority bit of the top bits -- the scalar unit has no 64-bit unsigned compare, and a compare on the
        // vector unit with a readfirstlane back would sit on the chain)
        const uint64_t carry = ((lo & inc_lo) | ((lo | inc_lo) & ~lo2)) >> 63;
        s_hi = hi + inc_hi + carry;
        s_lo = lo2;
    }
    __device__ __forceinline__ uint64_t output() const
    {
        const uint64_t x = s_hi ^ s_lo;
        const uint32_t rot = (uint32_t)(s_hi >> 58);
        return (x >> rot) | (x << ((64u - rot) & 63u));
    }
    __device__ __forceinline__ uint64_t next64() { advance(); return output(); }
    __device__ __forceinline__ uint32_t next32()
    {
        if (has_uint32) {
            has_uint32 = 0;
            return uinteger;
        }
        const uint64_t n = next64();
        has_uint32 = 1;
        uinteger = (uint32_t)(n >> 32);
        return (uint32_t)n;
    }
    __device__ __forceinline__ uint32_t below(uint32_t k)   // Generator.integers(0, k): as Pcg64::below
    {
        if (k <= 1) return 0;
        uint64_t m = (uint64_t)next32() * k;
        uint32_t leftover = (uint32_t)m;
        if (leftover < k) {
            const uint32_t threshold = (uint32_t)(0u - k) % k;
            while (leftover < threshold) {
                m = (uint64_t)next32() * k;
                leftover = (uint32_t)m;
            }
        }
        return (uint32_t)(m >> 32);
    }
};

} // namespace mp
