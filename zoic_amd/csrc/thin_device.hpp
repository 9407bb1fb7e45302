// thin_device.hpp -- one THINLENS camera ray in the reference's arithmetic (zoic.cpp:1771-1846), shared by the streaming
// kernel (kernels.hip) and the per-sample mailbox kernel (mailbox.hip).  STRICT: bit-exact against the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>

#include "device_search.hpp"
#include "optics.hpp"

#pragma STDC FP_CONTRACT OFF

namespace zoic {

// rowCellsLds: the workgroup's LDS copy of the bokeh row cell records (nullptr: pyramid / reference search)
__device__ __forceinline__ V2 sample_lens(bool useImage, const BokehTables &B, const float *rowCellsLds, int bw, int bh, float u, float v)
{
    if (useImage) {
        if (rowCellsLds) return bokeh_sample_cells<true>(B, rowCellsLds, bw, bh, u, v);
        return bokeh_sample_device(B, bw, bh, u, v);
    }
    return concentric_disk(u, v);
}

// empericalOpticalVignetting, zoic.cpp:1297-1305
__device__ __forceinline__ bool optical_vignet_pass(const ThinTable &T, V3 origin, V3 dir)
{
    const V3 p{dir.x * T.ovDistance - origin.x, dir.y * T.ovDistance - origin.y, dir.z * T.ovDistance - origin.z};
    const float hyp = sqrtf((p.x * p.x) + (p.y * p.y));
    return fabsf(hyp) < T.apertureRadius * T.ovRadius;
}

struct ThinRay { V3 origin, dir; float w; uint32_t tries; };

// s = (sx, sy, lensx, lensy); `seedStream` is called once, before the first redraw (zoic.cpp:1806), and must leave the ray's
// retry stream in `rng`
template <class SeedFn>
__device__ __forceinline__ ThinRay thin_ray_strict(const ThinTable &T, const BokehTables &B, const float *rowCells, float4 s, Rng &rng,
                                                   SeedFn seedStream)
{
    const bool useImage = T.useImage != 0;
    bool seeded = false;
    const V3 p{s.x * T.tanFov, s.y * T.tanFov, 1.0f};
    const V3 originOriginal{0.0f, 0.0f, 0.0f};  // Arnold hands output.origin in as 0 (zoic.cpp:1777 reads it)
    const V3 dir0 = normalize3(V3{p.x - originOriginal.x, p.y - originOriginal.y, p.z - originOriginal.z});
    ThinRay r;
    r.origin = originOriginal; r.dir = dir0; r.tries = 0; r.w = 1.0f;
    if (T.useDof) {
        V2 lens = sample_lens(useImage, B, rowCells, T.bokehW, T.bokehH, s.z, s.w);
        lens.x *= T.apertureRadius; lens.y *= T.apertureRadius;
        r.origin = V3{lens.x, lens.y, 0.0f};
        const float inter = fabsf(T.focalDistance / dir0.z);
        const V3 fp{dir0.x * inter, dir0.y * inter, dir0.z * inter};
        r.dir = normalize3(V3{fp.x - r.origin.x, fp.y - r.origin.y, fp.z - r.origin.z});
        if (T.ovDistance > 0.0f) {
            while (!optical_vignet_pass(T, r.origin, r.dir) && r.tries <= static_cast<uint32_t>(kMaxTries)) {  // zoic.cpp:1804-1819
                if (!seeded) { seedStream(); seeded = true; }
                const float u = rng_unit(xor128(rng));
                const float v = rng_unit(xor128(rng));
                lens = sample_lens(useImage, B, rowCells, T.bokehW, T.bokehH, u, v);
                lens.x *= T.apertureRadius; lens.y *= T.apertureRadius;
                r.origin = V3{lens.x, lens.y, 0.0f};
                r.dir = normalize3(V3{fp.x - r.origin.x, fp.y - r.origin.y, fp.z - r.origin.z});  // dir0, inter, fp are loop invariant
                ++r.tries;
            }
        }
        if (r.tries > static_cast<uint32_t>(kMaxTries)) r.w = 0.0f;   // zoic.cpp:1824-1830
    }
    r.dir.z = r.dir.z * -1.0f;                     // zoic.cpp:1845
    if (T.exposureOn) r.w *= T.exposureMul;
    return r;
}

}  // namespace zoic
