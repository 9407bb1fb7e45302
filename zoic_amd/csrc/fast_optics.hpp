// fast_optics.hpp -- FAST-mode device arithmetic of the Kolb path (gfx950).  Same algorithm and the same
// accept/reject formulas as optics.hpp (zoic.cpp:973-1158, 1850-1948), re-associated for the VALU:
//   * f32 only (the reference's f64 intermediates dropped), FMA contraction on;
//   * the ray direction is normalised once (v_rsq_f32) and then stays unit by construction -- the reference
//     re-normalises it at every surface and twice more inside Snell (zoic.cpp:974,1002,1009-1010);
//   * the surface normal is (centre - hit) * (1/R): |centre - hit| == |R| on the sphere, no sqrt;
//   * the cosine of incidence is thc/|R| (no dot product) and every factor of thc, thc^2 is folded into per-surface
//     constants on the host (tables.hpp FastSurface);
//   * v_sqrt_f32 (1 ulp) for the two remaining roots per surface; the next surface's constants are fetched with
//     scalar loads while the current surface is evaluated.
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

__device__ __forceinline__ float uniform_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ FastSurface uniform_surface(const FastSurface &s)
{
    FastSurface r;
    r.center = uniform_f32(s.center); r.radius2 = uniform_f32(s.radius2); r.sign = uniform_f32(s.sign);
    r.housing2 = uniform_f32(s.housing2); r.invRadius = uniform_f32(s.invRadius); r.eta = uniform_f32(s.eta);
    r.etaInvAbsR = uniform_f32(s.etaInvAbsR); r.e2InvR2 = uniform_f32(s.e2InvR2); r.oneMinusEta2 = uniform_f32(s.oneMinusEta2);
    r.pad0 = r.pad1 = r.pad2 = 0.0f;
    return r;
}

// One ray through the lens, rear -> front (traceThroughLensElements, zoic.cpp:1099-1158).
//   o : ray origin, advanced to the last accepted hit point (zoic.cpp:1130)
//   d : raw direction on entry; the refracted unit direction on exit.  If the ray dies before its first refraction d is
//       left untouched -- the partial state the reference hands out for rays that run out of tries (zoic.cpp:1951-1961).
// With |u| = 1 and the hit on the sphere:  N = (c - hit)/R,  cos(i) = -(u.N) = thc/|R|  (no dot product),
//   1 - cs2 = (1 - eta^2) + (eta/R)^2 thc^2,   TIR <=> 1 - cs2 < 0,
//   u' = eta u + (eta cos(i) - sqrt(1 - cs2)) N      (|u'| = 1 again).
// ~28 VALU + 2 v_sqrt_f32 per interface instead of ~52.
__device__ __forceinline__ bool trace_lens_fast(const KolbTable &T, V3 &o, V3 &d, uint32_t &tirCount)
{
    const int n = T.lensCount;
    const float inv = frsq_fast(d.x * d.x + d.y * d.y + d.z * d.z);
    V3 u{d.x * inv, d.y * inv, d.z * inv};
    bool ok = true, refracted = false;
    // Every lane still in the loop is at the same surface, but the divergent exits hide that from the compiler, which
    // would fetch the table per lane with vector loads and keep it in VGPRs.  readfirstlane pins the index and the
    // table words to SGPRs: scalar loads, issued one iteration ahead of their use.
    FastSurface Snext = T.fsurf[0];
    for (int i = 0;;) {
        const int iu = __builtin_amdgcn_readfirstlane(i);
        const FastSurface S = uniform_surface(Snext);
        Snext = T.fsurf[(iu + 1 < n) ? iu + 1 : iu];
        const float Lz = S.center - o.z;
        const float tca = Lz * u.z - o.x * u.x - o.y * u.y;
        const float d2 = (o.x * o.x + o.y * o.y + Lz * Lz) - tca * tca;
        const float w = fabsf(S.radius2 - d2);                 // thc^2
        const float thc = fsqrt_fast(w);
        const float t = tca + thc * S.sign;
        const V3 hit{o.x + u.x * t, o.y + u.y * t, o.z + u.z * t};
        const float h2 = hit.x * hit.x + hit.y * hit.y;
        const bool clipped = (d2 > S.radius2) | (h2 > S.housing2) | ((iu == T.apertureElement) & (h2 > T.userAperture2));
        if (clipped) { ok = false; break; }
        o = hit;
        const float oneMinusCs2 = S.oneMinusEta2 + S.e2InvR2 * w;
        if (oneMinusCs2 < 0.0f) { ++tirCount; ok = false; break; }   // cs2 > 1 (only reachable when eta > 1)
        const float k = thc * S.etaInvAbsR - fsqrt_fast(oneMinusCs2);
        const float kr = k * S.invRadius;                       // k * N = kr * (c - hit)
        u = V3{u.x * S.eta - hit.x * kr, u.y * S.eta - hit.y * kr, u.z * S.eta + (S.center - hit.z) * kr};
        refracted = true;
        if (++i == n) break;
    }
    if (refracted) d = u;
    return ok;
}

}  // namespace zoic

#pragma clang fp contract(off)
