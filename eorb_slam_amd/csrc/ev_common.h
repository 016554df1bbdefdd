// ev_common.h -- small device helpers shared by the accumulation translation units (ev_accum.hip, ev_slots.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eorb {

__device__ __forceinline__ uint32_t enc_f32(float f)
{   // order-preserving map float -> uint32 (for atomicMax/atomicMin on floats of either sign)
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t e)
{
    uint32_t u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
    return __uint_as_float(u);
}

// wave-wide inclusive prefix sum without LDS round trips: row_shr 1/2/4/8 inside the 16-lane rows, then row_bcast 15 / 31
__device__ __forceinline__ int wave_incl_scan(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);     // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);     // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);     // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);     // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return x;
}

// wave-wide maximum / minimum of a float by the same DPP steps (a lane without a source keeps its own value): the result is in lane 63
__device__ __forceinline__ float wave_max_to_lane63(float x)
{
#define EORB_DPP_STEP(ctrl, rmask) x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), ctrl, rmask, 0xf, false)))
    EORB_DPP_STEP(0x111, 0xf); EORB_DPP_STEP(0x112, 0xf); EORB_DPP_STEP(0x114, 0xf); EORB_DPP_STEP(0x118, 0xf);
    EORB_DPP_STEP(0x142, 0xa); EORB_DPP_STEP(0x143, 0xc);
#undef EORB_DPP_STEP
    return x;
}
__device__ __forceinline__ float wave_min_to_lane63(float x)
{
#define EORB_DPP_STEP(ctrl, rmask) x = fminf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), ctrl, rmask, 0xf, false)))
    EORB_DPP_STEP(0x111, 0xf); EORB_DPP_STEP(0x112, 0xf); EORB_DPP_STEP(0x114, 0xf); EORB_DPP_STEP(0x118, 0xf);
    EORB_DPP_STEP(0x142, 0xa); EORB_DPP_STEP(0x143, 0xc);
#undef EORB_DPP_STEP
    return x;
}

}  // namespace eorb
