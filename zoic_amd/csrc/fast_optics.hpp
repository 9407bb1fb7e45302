// fast_optics.hpp -- FAST-mode device arithmetic of the Kolb path (gfx950).  Same algorithm and the same
// accept/reject formulas as optics.hpp (zoic.cpp:973-1158, 1850-1948), re-associated for the VALU:
//   * f32 only (the reference's f64 intermediates dropped), FMA contraction on;
//   * the ray direction is normalised once (v_rsq_f32) and then stays unit by construction -- the reference
//     re-normalises it at every surface and twice more inside Snell (zoic.cpp:974,1002,1009-1010);
//   * the surface normal is (centre - hit) * (1/R): |centre - hit| == |R| on the sphere, no sqrt;
//   * v_sqrt_f32 (1 ulp) for the two remaining roots per surface; the next surface's constants are fetched
//     (one s_load_dwordx8) while the current surface is evaluated.
// Decisions flip only where the reference's own f32 rounding noise decides; measured in tests/test_parity_gpu.py.
#pragma once
#include <hip/hip_runtime.h>

#include "optics.hpp"

#pragma clang fp contract(fast)

namespace zoic {

__device__ __forceinline__ float fsqrt_fast(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float frsq_fast(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_sin_f32(float x) { return parabola_sin(wrap_to_pi(x)); }
__device__ __forceinline__ float fast_cos_f32(float x) { return parabola_sin(wrap_to_pi(x + kPiOver2)); }

__device__ __forceinline__ V2 concentric_disk_f32(float ox, float oy)
{
    const float a = 2.0f * ox - 1.0f, b = 2.0f * oy - 1.0f;
    float r, phi;
    if ((a * a) > (b * b)) { r = a; phi = 0.78539816339f * (b / a); }
    else { r = b; phi = kPiOver2 - 0.78539816339f * (a / b); }
    return V2{r * fast_cos_f32(phi), r * fast_sin_f32(phi)};
}

// d: raw (unnormalised) direction on entry; replaced by the refracted unit direction at the first surface and left
// untouched if the ray dies before that -- the partial state the reference leaves behind (zoic.cpp:1951-1961).
__device__ __forceinline__ bool trace_lens_fast(const KolbTable &T, V3 &o, V3 &d, uint32_t &tirCount)
{
    const int n = T.lensCount;
    const float inv = frsq_fast(d.x * d.x + d.y * d.y + d.z * d.z);
    V3 u{d.x * inv, d.y * inv, d.z * inv};
    bool ok = true;
    Surface S = T.surf[0];
    for (int i = 0;;) {
        const Surface Sn = T.surf[(i + 1 < n) ? i + 1 : i];  // scalar prefetch of the next interface
        const float Lx = -o.x, Ly = -o.y, Lz = S.center - o.z;
        const float tca = Lx * u.x + Ly * u.y + Lz * u.z;
        const float d2 = (Lx * Lx + Ly * Ly + Lz * Lz) - tca * tca;
        const float thc = fsqrt_fast(fabsf(S.radius2 - d2));
        const float t = tca + thc * S.sign;
        const V3 hit{o.x + u.x * t, o.y + u.y * t, o.z + u.z * t};
        const float h2 = hit.x * hit.x + hit.y * hit.y;
        const bool clipped = (d2 > S.radius2) | (h2 > S.housing2) | ((i == T.apertureElement) & (h2 > T.userAperture2));
        if (clipped) { ok = false; break; }
        o = hit;
        const V3 N{-hit.x * S.invRadius, -hit.y * S.invRadius, (S.center - hit.z) * S.invRadius};
        const float c1 = -(u.x * N.x + u.y * N.y + u.z * N.z);
        const float cs2 = (S.eta * S.eta) * (1.0f - c1 * c1);
        if (S.tirPossible && cs2 > 1.0f) { ++tirCount; ok = false; break; }
        const float k = S.eta * c1 - fsqrt_fast(fabsf(1.0f - cs2));
        u = V3{u.x * S.eta + N.x * k, u.y * S.eta + N.y * k, u.z * S.eta + N.z * k};
        d = u;
        if (++i == n) break;
        S = Sn;
    }
    return ok;
}

}  // namespace zoic

#pragma clang fp contract(off)
