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

__device__ __forceinline__ float frcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }

// concentricDiskSample (zoic.cpp:686-704) with one v_rcp_f32 instead of an IEEE divide; branch-free
__device__ __forceinline__ V2 concentric_disk_f32(float ox, float oy)
{
    const float a = 2.0f * ox - 1.0f, b = 2.0f * oy - 1.0f;
    const bool wide = (a * a) > (b * b);
    const float num = wide ? b : a, den = wide ? a : b;
    const float q = 0.78539816339f * (num * frcp_fast(den));   // 0/0 -> NaN like the reference
    const float phi = wide ? q : kPiOver2 - q;
    return V2{den * fast_cos_f32(phi), den * fast_sin_f32(phi)};
}

// atan2 to ~2e-7 rad (the parabola sin/cos that consume it are 1e-3 approximations of sin/cos anyway, but they must
// see the reference's angle): minimax odd polynomial on [0,1] + octant unfolding, one v_rcp_f32, no branches.
__device__ __forceinline__ float atan2_f32(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * frcp_fast(mx);            // in [0,1]; 0/0 -> NaN -> handled below
    const float s = t * t;
    // atan(t)/t on [0,1], degree-7 in s (Abramowitz-Stegun style minimax, max error ~1e-7)
    float p = -0.0040540580f;
    p = p * s + 0.0218612288f;
    p = p * s - 0.0559098861f;
    p = p * s + 0.0964200441f;
    p = p * s - 0.1390853351f;
    p = p * s + 0.1994653599f;
    p = p * s - 0.3332985605f;
    p = p * s + 0.9999993329f;
    float r = p * t;
    r = (ay > ax) ? kPiOver2 - r : r;
    r = (x < 0.0f) ? kPi - r : r;
    r = (mx == 0.0f) ? 0.0f : r;                   // atan2(0,0) = 0
    return copysignf(r, y);
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
//
// One interface of the fast trace.  Returns 0 = passed, 1 = clipped (o, u untouched), 2 = total internal reflection
// (o advanced, u untouched) -- the two partial states the reference can leave.
__device__ __forceinline__ int fast_interface(const FastSurface &S, bool isStop, float userAperture2, V3 &o, V3 &u)
{
    const float Lz = S.center - o.z;
    const float tca = Lz * u.z - o.x * u.x - o.y * u.y;
    const float d2 = (o.x * o.x + o.y * o.y + Lz * Lz) - tca * tca;
    const float w = fabsf(S.radius2 - d2);                 // thc^2
    const float thc = fsqrt_fast(w);
    const float t = tca + thc * S.sign;
    const V3 hit{o.x + u.x * t, o.y + u.y * t, o.z + u.z * t};
    const float h2 = hit.x * hit.x + hit.y * hit.y;
    const bool clipped = (d2 > S.radius2) | (h2 > S.housing2);   // the stop's housing2 includes the user aperture
    if (clipped) return 1;
    o = hit;
    const float oneMinusCs2 = S.oneMinusEta2 + S.e2InvR2 * w;
    if (oneMinusCs2 < 0.0f) return 2;                       // cs2 > 1 (only reachable when eta > 1)
    const float k = thc * S.etaInvAbsR - fsqrt_fast(oneMinusCs2);
    const float kr = k * S.invRadius;                       // k * N = kr * (c - hit)
    u = V3{u.x * S.eta - hit.x * kr, u.y * S.eta - hit.y * kr, u.z * S.eta + (S.center - hit.z) * kr};
    return 0;
}

// Does a ray clear interface 0 (first sphere hit + rear-element housing)?  Same arithmetic as the first half of
// fast_interface; used by the kernel's candidate search so that tries dying at the rear element never pay for a trace.
__device__ __forceinline__ bool interface0_clear_fast(const FastSurface &S, V3 o, V3 d)
{
    const float inv = frsq_fast(d.x * d.x + d.y * d.y + d.z * d.z);
    const V3 u{d.x * inv, d.y * inv, d.z * inv};
    const float Lz = S.center - o.z;
    const float tca = Lz * u.z - o.x * u.x - o.y * u.y;
    const float d2 = (o.x * o.x + o.y * o.y + Lz * Lz) - tca * tca;
    const float thc = fsqrt_fast(fabsf(S.radius2 - d2));
    const float t = tca + thc * S.sign;
    const float hx = o.x + u.x * t, hy = o.y + u.y * t;
    const float h2 = hx * hx + hy * hy;
    return !((d2 > S.radius2) | (h2 > S.housing2));
}

// Rolled, branchy trace for any interface count.  It leaves exactly the partial state of the reference on every exit
// path, so it is also what finishes rays that ran out of tries in the predicated kernel below.
__device__ __forceinline__ bool trace_lens_fast_rolled(const KolbTable &T, V3 &o, V3 &d, uint32_t &tirCount)
{
    const float inv = frsq_fast(d.x * d.x + d.y * d.y + d.z * d.z);
    V3 u{d.x * inv, d.y * inv, d.z * inv};
    bool ok = true, refracted = false;
    const int n = T.lensCount;
    // Every lane still in the loop is at the same surface, but the divergent exits hide that from the compiler;
    // readfirstlane pins the index and the table words to SGPRs (scalar loads, one iteration ahead).
    FastSurface Snext = T.fsurf[0];
    for (int i = 0;;) {
        const int iu = __builtin_amdgcn_readfirstlane(i);
        const FastSurface S = uniform_surface(Snext);
        Snext = T.fsurf[(iu + 1 < n) ? iu + 1 : iu];
        const int r = fast_interface(S, iu == T.apertureElement, T.userAperture2, o, u);
        if (r != 0) { if (r == 2) ++tirCount; ok = false; break; }
        refracted = true;
        if (++i == n) break;
    }
    if (refracted) d = u;
    return ok;
}

// Predicated, fully unrolled trace for a lens with exactly NS interfaces: NO divergent control flow.  Every lane
// evaluates every interface; a lane that is clipped or totally reflected only clears its bit in the `alive` mask
// (v_cmp -> SGPR pair, s_andn2).  Dead lanes keep computing on garbage, which costs nothing: a wave issues each VALU
// instruction once whatever its exec mask.  The rolled loop above spends ~33 scalar instructions per interface on
// exec-mask bookkeeping and loop control next to 37 VALU, and the scalar unit became the limiter; here an interface
// is ~31 VALU + ~6 SALU, and its table words are s_loads at fixed kernel-argument offsets (SGPR operands: staging
// the table in LDS instead was measured 36 % slower on C4 -- VGPR copies, no scalar operands).  A wave-uniform test
// every second interface leaves the trace as soon as no lane is alive (heavily vignetted passes).
// Returns alive; o/u are the exit point and unit direction for alive lanes (unspecified for dead ones -- rays that
// finish dead get their reference partial state from trace_lens_fast_rolled).
template <int NS>
__device__ __forceinline__ bool trace_lens_fast_pred(const FastSurface *__restrict__ surf, V3 &o, V3 &d, uint32_t &tirCount, bool alive0)
{
    static_assert(NS > 0, "predicated trace needs a compile-time interface count");
    const float inv = frsq_fast(d.x * d.x + d.y * d.y + d.z * d.z);
    V3 u{d.x * inv, d.y * inv, d.z * inv};
    bool alive = alive0, tirSeen = false;   // alive0: lanes without a candidate ride along dead
    bool anyAlive = true;  // wave-uniform
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (i >= 2 && (i & 1) == 0) anyAlive = __ballot(alive) != 0ull;   // wave-uniform early out, every 2nd interface
        if (!anyAlive) continue;
        const FastSurface S = surf[i];
        const float Lz = S.center - o.z;
        const float tca = Lz * u.z - o.x * u.x - o.y * u.y;
        const float d2 = (o.x * o.x + o.y * o.y + Lz * Lz) - tca * tca;
        const float w = fabsf(S.radius2 - d2);                 // thc^2
        const float thc = fsqrt_fast(w);
        const float t = tca + thc * S.sign;
        o = V3{o.x + u.x * t, o.y + u.y * t, o.z + u.z * t};    // hit point
        const float h2 = o.x * o.x + o.y * o.y;
        const bool clipped = (d2 > S.radius2) | (h2 > S.housing2);   // the stop's housing2 includes the user aperture
        const float oneMinusCs2 = S.oneMinusEta2 + S.e2InvR2 * w;
        const bool tirHere = oneMinusCs2 < 0.0f;
        tirSeen |= alive & !clipped & tirHere;                  // counted only by rays that reached the refraction
        alive &= !clipped & !tirHere;
        const float k = thc * S.etaInvAbsR - fsqrt_fast(fabsf(oneMinusCs2));
        const float kr = k * S.invRadius;
        u = V3{u.x * S.eta - o.x * kr, u.y * S.eta - o.y * kr, u.z * S.eta + (S.center - o.z) * kr};
    }
    tirCount += tirSeen ? 1u : 0u;
    d = u;
    return alive;
}

}  // namespace zoic

#pragma clang fp contract(off)
