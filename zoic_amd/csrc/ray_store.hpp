// ray_store.hpp -- device-side store of one finished ray (kernels.hpp: RayRecord).  Included by .hip files only.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace zoic {

// two aligned 16-byte stores = one whole 32-byte sector per lane
__device__ __forceinline__ void store_ray_record(RayRecord *out, uint64_t i, float ox, float oy, float oz, float dx, float dy,
                                                 float dz, float w, uint32_t flags)
{
    float4 *p = reinterpret_cast<float4 *>(out + i);
    p[0] = make_float4(ox, oy, oz, dx);
    p[1] = make_float4(dy, dz, w, __builtin_bit_cast(float, flags));
}

// Wave-cooperative store for kernels where lane l of a wave owns ray (waveBase + l): the 64 records (2 KiB) are
// transposed through LDS so that each of the two store instructions writes one contiguous, fully coalesced KiB
// (lane l writes 16-byte piece k*64 + l) instead of 64 half-sectors at a 32-byte stride.  `stage` = 128 float4 of
// LDS private to the wave; `valid` = number of rays of this wave that exist (ragged last tile).
__device__ __forceinline__ void store_ray_records_wave(RayRecord *out, uint64_t waveBase, uint32_t lane, uint32_t valid, float4 *stage,
                                                       float ox, float oy, float oz, float dx, float dy, float dz, float w,
                                                       uint32_t flags)
{
    stage[2 * lane] = make_float4(ox, oy, oz, dx);
    stage[2 * lane + 1] = make_float4(dy, dz, w, __builtin_bit_cast(float, flags));
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's LDS writes have landed (same-wave LDS ops are ordered)
    float4 *dst = reinterpret_cast<float4 *>(out + waveBase);
    const float4 a = stage[lane], b = stage[64 + lane];
    if (lane < 2 * valid) dst[lane] = a;
    if (64 + lane < 2 * valid) dst[64 + lane] = b;
    __builtin_amdgcn_wave_barrier();
}

}  // namespace zoic
