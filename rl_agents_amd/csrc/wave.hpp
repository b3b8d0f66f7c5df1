// wave.hpp -- cross-lane reductions on a 64-lane wavefront with DPP (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace mp {

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

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int min_step(int v)
{
    const int o = __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
    return o < v ? o : v;
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
    v = min_step<0x111, 0xf>(v);
    v = min_step<0x112, 0xf>(v);
    v = min_step<0x114, 0xf>(v);
    v = min_step<0x118, 0xf>(v);
    v = min_step<0x142, 0xa>(v);
    v = min_step<0x143, 0xc>(v);
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int row0_min(int v)
{
    v = min_step<0x111, 0xf>(v);
    v = min_step<0x112, 0xf>(v);
    v = min_step<0x114, 0xf>(v);
    v = min_step<0x118, 0xf>(v);
    return __builtin_amdgcn_readlane(v, 15);
}

// ---- cross-lane argmax on (U, id): maximal U first, lowest id among equal U; every lane returns the pair.
// Two separable reductions -- the maximum of U (3 instructions per step), then the minimum id among the lanes that hold
// it (2 per step) -- instead of one lexicographic reduction whose compare-and-select on a 96-bit pair costs ~28
// instructions per step: a planner wave is ONE instruction stream, and the two argmaxes of an OPD expansion were
// two thirds of the instructions it issued.
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
