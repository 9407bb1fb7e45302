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

#include <cstddef>

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

// The per-surface constants are wave-uniform kernel arguments: read through a pointer in the CONSTANT address space they
// stay s_loads (SGPR operands).  The persistent pass loop re-derives this pointer every pass behind an empty asm
// (launder_table): otherwise LLVM hoists all NS x 10 loads out of the pass loop as loop invariants, runs out of SGPRs and
// spills them to VGPR lanes -- every constant then costs a v_readlane per use (measured: 568 v_readlane + VGPR scratch
// spills in the decision-safe kernel, -25 % throughput) instead of a scalar-cache hit.
#ifndef ZOIC_GUARD_PIN
#define ZOIC_GUARD_PIN 1
#endif
typedef const FastSurface __attribute__((address_space(4))) *FastSurfaceTable;
// The KolbTable is the FIRST kernel argument of every Kolb kernel (offset 0 of the kernarg segment): its FastSurface array
// is addressed from the kernarg base.  (Deriving the pointer from &T instead makes the by-value argument's address escape
// and LLVM copies parts of it to scratch.)
__device__ __forceinline__ FastSurfaceTable kernarg_fast_surfaces()
{
    typedef const char __attribute__((address_space(4))) *KernargBytes;
    return (FastSurfaceTable)((KernargBytes)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KolbTable, fsurf));
}
__device__ __forceinline__ FastSurfaceTable launder_table(FastSurfaceTable t)
{
    asm volatile("" : "+s"(t));
    return t;
}
template <bool PIN = true>
__device__ __forceinline__ FastSurface load_surface(FastSurfaceTable t, int i)
{
    if constexpr (PIN) asm volatile("" : "+s"(t));   // the loads of interface i are issued here, not hoisted to the top of the pass
    FastSurface S;
    S.center = t[i].center; S.radius2 = t[i].radius2; S.sign = t[i].sign; S.housing2 = t[i].housing2; S.invRadius = t[i].invRadius;
    S.eta = t[i].eta; S.etaInvAbsR = t[i].etaInvAbsR; S.e2InvR2 = t[i].e2InvR2; S.oneMinusEta2 = t[i].oneMinusEta2;
    S.bandHousing = t[i].bandHousing; S.pad0 = S.pad1 = 0.0f;
    return S;
}

// Decision-safe mode: only ill-conditioned interfaces carry a guard band (bandHousing > 0, set by the host: in practice the
// stop, see lens_system.cpp fill_surfaces); the test is a wave-uniform scalar branch, so well-conditioned interfaces pay nothing.
__device__ __forceinline__ bool surface_is_guarded(const FastSurface &S) { return __builtin_bit_cast(int, S.bandHousing) > 0; }

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
    r.bandHousing = uniform_f32(s.bandHousing); r.pad0 = r.pad1 = 0.0f;
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
__device__ __forceinline__ int fast_interface(const FastSurface &S, bool isStop, float userAperture2, V3 &o, V3 &u, bool *near = nullptr)
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
    const float oneMinusCs2 = S.oneMinusEta2 + S.e2InvR2 * w;
    if (near) *near = surface_is_guarded(S) && fabsf(h2 - S.housing2) < S.bandHousing;
    if (clipped) return 1;
    o = hit;
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

// The same test for the decision-safe mode: `near` is set when the housing decision lies inside its guard band (tables.hpp).
__device__ __forceinline__ bool interface0_clear_fast_guard(const FastSurface &S, V3 o, V3 d, bool &near)
{
    const float inv = frsq_fast(d.x * d.x + d.y * d.y + d.z * d.z);
    const V3 u{d.x * inv, d.y * inv, d.z * inv};
    const float Lz = S.center - o.z;
    const float tca = Lz * u.z - o.x * u.x - o.y * u.y;
    const float d2 = (o.x * o.x + o.y * o.y + Lz * Lz) - tca * tca;
    const float w = fabsf(S.radius2 - d2);
    const float thc = fsqrt_fast(w);
    const float t = tca + thc * S.sign;
    const float hx = o.x + u.x * t, hy = o.y + u.y * t;
    const float h2 = hx * hx + hy * hy;
    near = fabsf(h2 - S.housing2) < S.bandHousing;
    return !((d2 > S.radius2) | (h2 > S.housing2));
}

// Rolled, branchy trace for any interface count.  It leaves exactly the partial state of the reference on every exit
// path, so it is also what finishes rays that ran out of tries in the predicated kernel below.
__device__ __forceinline__ bool trace_lens_fast_rolled(const KolbTable &T, V3 &o, V3 &d, uint32_t &tirCount, bool *unsure = nullptr)
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
        bool near = false;
        const int r = fast_interface(S, iu == T.apertureElement, T.userAperture2, o, u, unsure ? &near : nullptr);
        if (unsure) *unsure |= near;
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
// GUARD (decision-safe mode): `unsure` collects, for lanes still alive at a guarded interface, whether its clip decision
// lies inside the guard band.
template <int NS, bool GUARD = false>
__device__ __forceinline__ bool trace_lens_fast_pred(FastSurfaceTable surf, V3 &o, V3 &d, uint32_t &tirCount, bool alive0,
                                                     bool *unsureOut = nullptr)
{
    static_assert(NS > 0, "predicated trace needs a compile-time interface count");
    const float inv = frsq_fast(d.x * d.x + d.y * d.y + d.z * d.z);
    V3 u{d.x * inv, d.y * inv, d.z * inv};
    bool alive = alive0, tirSeen = false;   // alive0: lanes without a candidate ride along dead
    bool unsure = false;
    bool anyAlive = true;  // wave-uniform
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (i >= 2 && (i & 1) == 0) anyAlive = __ballot(alive) != 0ull;   // wave-uniform early out, every 2nd interface
        if (!anyAlive) continue;
        const FastSurface S = load_surface<GUARD && (ZOIC_GUARD_PIN != 0)>(surf, i);
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
        if constexpr (GUARD) {
            unsure |= alive & (fabsf(h2 - S.housing2) < S.bandHousing);   // never true on unguarded interfaces (band 0): no branch
        }
        tirSeen |= alive & !clipped & tirHere;                  // counted only by rays that reached the refraction
        alive &= !clipped & !tirHere;
        const float k = thc * S.etaInvAbsR - fsqrt_fast(fabsf(oneMinusCs2));
        const float kr = k * S.invRadius;
        u = V3{u.x * S.eta - o.x * kr, u.y * S.eta - o.y * kr, u.z * S.eta + (S.center - o.z) * kr};
    }
    tirCount += tirSeen ? 1u : 0u;
    d = u;
    if constexpr (GUARD) *unsureOut = unsure;
    return alive;
}

}  // namespace zoic

#pragma clang fp contract(off)
