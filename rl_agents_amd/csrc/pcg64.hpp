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
    __device__ __forceinline__ uint64_t next64()
    {
        // state * 0x2360ED051FC65DA44385DF649FCCF645 + inc  (mod 2^128), schoolbook on 32-bit limbs:
        // the 6 partial products that feed carries as 32x32+64 multiply-adds, the 4 of the top limb as
        // low-half multiplies -- 10 multiplies (the generic 64-bit formulation compiles to 18).
        const uint32_t m0 = 0x9FCCF645u, m1 = 0x4385DF64u, m2 = 0x1FC65DA4u, m3 = 0x2360ED05u;
        const uint32_t a0 = (uint32_t)s_lo, a1 = (uint32_t)(s_lo >> 32), a2 = (uint32_t)s_hi, a3 = (uint32_t)(s_hi >> 32);
        const uint64_t c0 = (uint64_t)a0 * m0;
        const uint64_t t1 = (uint64_t)a0 * m1 + (c0 >> 32);
        const uint64_t u1 = (uint64_t)a1 * m0 + (uint32_t)t1;
        const uint64_t k2 = (t1 >> 32) + (u1 >> 32);
        const uint64_t v2 = (uint64_t)a0 * m2 + k2;
        const uint64_t w2 = (uint64_t)a1 * m1 + (uint32_t)v2;
        const uint64_t x2 = (uint64_t)a2 * m0 + (uint32_t)w2;
        const uint32_t k3 = (uint32_t)(v2 >> 32) + (uint32_t)(w2 >> 32) + (uint32_t)(x2 >> 32);
        const uint32_t r3 = a0 * m3 + a1 * m2 + a2 * m1 + a3 * m0 + k3;
        const uint64_t lo = (uint64_t)(uint32_t)c0 | ((uint64_t)(uint32_t)u1 << 32);
        uint64_t hi = (uint64_t)(uint32_t)x2 | ((uint64_t)r3 << 32);
        const uint64_t lo2 = lo + inc_lo;
        hi += inc_hi + (lo2 < lo ? 1ULL : 0ULL);
        s_hi = hi; s_lo = lo2;
        const uint64_t x = hi ^ lo2;
        const unsigned rot = (unsigned)(hi >> 58);
        return (x >> rot) | (x << ((64u - rot) & 63u));
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

} // namespace mp
