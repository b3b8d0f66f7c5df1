// wave.hpp -- cross-lane reductions on a 64-lane wavefront with DPP (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace mp {

// ---- cross-lane argmax on (U, id) with DPP (row_shr 1/2/4/8, row_bcast 15/31): VALU-speed data
// movement instead of 18 ds_bpermute round trips.  Maximal U first, lowest id among equal U.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void argmax_step(double &u, int &id)
{
    const int lo = __double2loint(u), hi = __double2hiint(u);
    // old = own value: lanes without a valid DPP source compare with themselves (no change)
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const int oid = __builtin_amdgcn_update_dpp(id, id, CTRL, ROW_MASK, 0xf, false);
    const double ou = __hiloint2double(ohi, olo);
    if (ou > u || (ou == u && oid < id)) { u = ou; id = oid; }
}

// every lane returns the wave-wide (max U, lowest id among maxima)
__device__ __forceinline__ void wave_argmax(double &u, int &id)
{
    argmax_step<0x111, 0xf>(u, id); // row_shr:1
    argmax_step<0x112, 0xf>(u, id); // row_shr:2
    argmax_step<0x114, 0xf>(u, id); // row_shr:4
    argmax_step<0x118, 0xf>(u, id); // row_shr:8   -> lane 15 of each row holds the row result
    argmax_step<0x142, 0xa>(u, id); // row_bcast:15 -> rows 1, 3
    argmax_step<0x143, 0xc>(u, id); // row_bcast:31 -> rows 2, 3: lane 63 holds the wave result
    const int lo = __builtin_amdgcn_readlane(__double2loint(u), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(u), 63);
    u = __hiloint2double(hi, lo);
    id = __builtin_amdgcn_readlane(id, 63);
}

} // namespace mp
