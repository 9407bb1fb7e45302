// ray_store.hpp -- device-side store of one finished ray (kernels.hpp: RayRecord).  Included by .hip files only.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace zoic {

__device__ __forceinline__ void nt_store(float4 *p, const float4 v)
{
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);   // one global_store_dwordx4 ... nt
}
__device__ __forceinline__ float4 nt_load(const float4 *p)
{
    return make_float4(__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y), __builtin_nontemporal_load(&p->z),
                       __builtin_nontemporal_load(&p->w));                               // one global_load_dwordx4 ... nt
}

// two aligned 16-byte stores = one whole 32-byte sector per lane
__device__ __forceinline__ void store_ray_record(RayRecord *out, uint64_t i, float ox, float oy, float oz, float dx, float dy,
                                                 float dz, float w, uint32_t flags)
{
    float4 *p = reinterpret_cast<float4 *>(out + i);
    p[0] = make_float4(ox, oy, oz, dx);
    p[1] = make_float4(dy, dz, w, __builtin_bit_cast(float, flags));
}

// the same with non-temporal stores (experiments: -DZOIC_STORE_NT, kolb_pool_body.hpp)
__device__ __forceinline__ void store_ray_record_nt(RayRecord *out, uint64_t i, float ox, float oy, float oz, float dx, float dy,
                                                    float dz, float w, uint32_t flags)
{
    float4 *p = reinterpret_cast<float4 *>(out + i);
    nt_store(p, make_float4(ox, oy, oz, dx));
    nt_store(p + 1, make_float4(dy, dz, w, __builtin_bit_cast(float, flags)));
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
    // streamed once, never re-read by this launch: non-temporal (thin lens 4.9 -> 5.3 TB/s)
    if (lane < 2 * valid) nt_store(dst + lane, a);
    if (64 + lane < 2 * valid) nt_store(dst + 64 + lane, b);

    __builtin_amdgcn_wave_barrier();
}

// LDS -> HBM for the records a wave parked in its last pass.  `stage` holds the 64 record slots as 128 consecutive
// 16-byte pieces, `stageIdx` the ray index per slot (0xffffffff: empty).  Lane j writes piece j, then piece 64 + j:
// neighbouring lanes write the two halves of one record, and rays refilled together (consecutive indices in lane order)
// mostly finish together, so one store instruction covers whole 32-byte sectors in long contiguous runs instead of 64
// half-sectors at a 32-byte stride.
__device__ __forceinline__ void flush_parked_records(RayRecord *out, const float4 *stage, const uint32_t *stageIdx, uint32_t lane)
{
#pragma unroll
    for (uint32_t h = 0; h < 2; ++h) {
        const uint32_t j = lane + 64u * h;
        const uint32_t id = stageIdx[j >> 1];
        const float4 piece = stage[j];
        if (id != 0xffffffffu) reinterpret_cast<float4 *>(out + id)[j & 1u] = piece;
    }
}

}  // namespace zoic
