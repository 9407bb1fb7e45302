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

}  // namespace zoic
