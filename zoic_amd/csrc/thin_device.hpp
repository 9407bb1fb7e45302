// thin_device.hpp -- one THINLENS camera ray in the reference's arithmetic (zoic.cpp:1771-1846), shared by the streaming
// kernel (kernels.hip) and the per-sample mailbox kernel (mailbox.hip).  STRICT: bit-exact against the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>

#include "device_search.hpp"
#include "fast_optics.hpp"
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

// ONE test of the retry loop (zoic.cpp:1804-1819) for the lens draw (u, v): the ray it makes and whether it clears the optical
// vignetting.  FAST (zoic_camera_set_precision, opticalVignettingDistance > 0): f32 rsq normalisation, the f32 disk mapping and
// v_sqrt in the vignetting test instead of the reference's correctly rounded divides and square roots -- ~80 instead of ~140
// instructions per redraw; direction error ~1e-7; a vignetting test within a few ulps of its limit is re-taken in the reference's
// arithmetic (decision-safe, like the Kolb kernels' guard band).  STRICT is bit-exact.  Shared by the batch kernel
// (thin_refill.hip) and the resident tile workers (mailbox.hip): a sample's ray is the same bits whichever serves it.
template <bool FAST>
__device__ __forceinline__ bool thin_vignet_try(const ThinTable &T, const BokehTables &B, const float *rowCells, bool useImage, float fpx,
                                                float fpy, float fpz, float u, float v, V3 &origin, V3 &dir)
{
    V2 lens = useImage ? (rowCells ? bokeh_sample_cells<!FAST>(B, rowCells, T.bokehW, T.bokehH, u, v)
                                   : bokeh_sample_device(B, T.bokehW, T.bokehH, u, v))
                       : (FAST ? concentric_disk_f32(u, v) : concentric_disk(u, v));
    lens.x *= T.apertureRadius; lens.y *= T.apertureRadius;
    origin = V3{lens.x, lens.y, 0.0f};
    bool clear;
    if constexpr (FAST) {
        const V3 q{fpx - origin.x, fpy - origin.y, fpz - origin.z};
        const float inv = frsq_fast(q.x * q.x + q.y * q.y + q.z * q.z);
        dir = V3{q.x * inv, q.y * inv, q.z * inv};
        const float ax = dir.x * T.ovDistance, ay = dir.y * T.ovDistance;
        const float px = ax - origin.x, py = ay - origin.y;
        const float hyp = fsqrt_fast(px * px + py * py), lim = T.apertureRadius * T.ovRadius;
        clear = hyp < lim;
        // Decision-safe: the f32 shortcuts above are good to a few ulps of the terms of p; a test that close to the
        // limit is re-taken in the reference's arithmetic, lens sample included (rare and divergent -- except on
        // degenerate settings such as a vignetting distance of ~0 behind an image whose rim pixels sit ON the limit).
        if (fabsf(hyp - lim) <= 2.0e-6f * (fabsf(ax) + fabsf(ay) + fabsf(origin.x) + fabsf(origin.y))) {
            V2 ls = useImage ? (rowCells ? bokeh_sample_cells<true>(B, rowCells, T.bokehW, T.bokehH, u, v)
                                         : bokeh_sample_device(B, T.bokehW, T.bokehH, u, v))
                             : concentric_disk(u, v);
            ls.x *= T.apertureRadius; ls.y *= T.apertureRadius;
            origin = V3{ls.x, ls.y, 0.0f};
            dir = normalize3(V3{fpx - origin.x, fpy - origin.y, fpz - origin.z});
            clear = optical_vignet_pass(T, origin, dir);
        }
    } else {
        dir = normalize3(V3{fpx - origin.x, fpy - origin.y, fpz - origin.z});
        clear = optical_vignet_pass(T, origin, dir);
    }
    return clear;
}

// THINLENS with optical vignetting on (useDof, opticalVignettingDistance > 0), one ray, FAST arithmetic: the loop
// thin_refill.hip's kernel runs a lane at a time -- same tests (thin_vignet_try<true>), same draws, same bits.
template <class SeedFn>
__device__ __forceinline__ ThinRay thin_ray_fast_vignet(const ThinTable &T, const BokehTables &B, const float *rowCells, float4 s, Rng &rng,
                                                        SeedFn seedStream)
{
    const bool useImage = T.useImage != 0;
    const V3 p{s.x * T.tanFov, s.y * T.tanFov, 1.0f};           // zoic.cpp:1773-1777 (output.origin arrives as 0)
    const V3 dir0 = normalize3(V3{p.x - 0.0f, p.y - 0.0f, p.z - 0.0f});
    const float inter = fabsf(T.focalDistance / dir0.z);        // zoic.cpp:1796-1797
    const float fpx = dir0.x * inter, fpy = dir0.y * inter, fpz = dir0.z * inter;
    ThinRay r;
    r.tries = 0;
    float u = s.z, v = s.w;
    for (;;) {
        const bool clear = thin_vignet_try<true>(T, B, rowCells, useImage, fpx, fpy, fpz, u, v, r.origin, r.dir);
        if (clear || r.tries > static_cast<uint32_t>(kMaxTries)) break;
        if (r.tries == 0) seedStream();
        u = rng_unit(xor128(rng));
        v = rng_unit(xor128(rng));
        ++r.tries;
    }
    r.w = (r.tries > static_cast<uint32_t>(kMaxTries)) ? 0.0f : 1.0f;      // zoic.cpp:1824-1830
    r.dir.z = r.dir.z * -1.0f;                                             // zoic.cpp:1845
    if (T.exposureOn) r.w *= T.exposureMul;
    return r;
}

}  // namespace zoic
