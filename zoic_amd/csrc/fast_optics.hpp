// fast_optics.hpp -- FAST-mode device arithmetic of the Kolb path (gfx950).  Same algorithm and the same
// accept/reject formulas as optics.hpp (zoic.cpp:973-1158, 1850-1948), re-associated for the VALU:
//   * f32 only (the reference's f64 intermediates dropped), with EXPLICIT fused multiply-adds (ffma below) and contraction OFF:
//     every rounding of the FAST arithmetic is written in this file.  (Rounds 1-3 let the compiler contract -- `fp contract(fast)`
//     -- and the same source then rounded differently in the rolled trace, the unrolled trace and in each kernel it was inlined
//     into: 0.7 % of the rays differed in their last bits between two code paths, which a ray that may be evaluated by either
//     of two kernels -- kolb_listed_body.hpp -- cannot afford: SURVEY 8e wants a sharded frame bit-identical to the one-GPU frame.)
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

#pragma clang fp contract(off)

namespace zoic {

__device__ __forceinline__ float ffma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }   // one v_fma_f32, one rounding
__device__ __forceinline__ float fsqrt_fast(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float frsq_fast(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_sin_f32(float x) { return parabola_sin(wrap_to_pi(x)); }
__device__ __forceinline__ float fast_cos_f32(float x) { return parabola_sin(wrap_to_pi(x + kPiOver2)); }

__device__ __forceinline__ float frcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }

// concentricDiskSample (zoic.cpp:686-704) with one v_rcp_f32 instead of an IEEE divide; branch-free
__device__ __forceinline__ V2 concentric_disk_f32(float ox, float oy)
{
    const float a = ffma(2.0f, ox, -1.0f), b = ffma(2.0f, oy, -1.0f);
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
    p = ffma(p, s, 0.0218612288f);
    p = ffma(p, s, -0.0559098861f);
    p = ffma(p, s, 0.0964200441f);
    p = ffma(p, s, -0.1390853351f);
    p = ffma(p, s, 0.1994653599f);
    p = ffma(p, s, -0.3332985605f);
    p = ffma(p, s, 0.9999993329f);
    float r = p * t;
    r = (ay > ax) ? kPiOver2 - r : r;
    r = (x < 0.0f) ? kPi - r : r;
    r = (mx == 0.0f) ? 0.0f : r;                   // atan2(0,0) = 0
    return copysignf(r, y);
}

// The per-surface constants are wave-uniform kernel arguments: read through a pointer in the CONSTANT address space they
// stay s_loads (SGPR operands).  The pass loop re-derives this pointer every pass behind an empty asm (launder_table):
// otherwise LLVM hoists all NS x 10 loads out of the pass loop as loop invariants, runs out of SGPRs and spills them to
// VGPR lanes -- every constant then costs a v_readlane per use (measured: 568 v_readlane + VGPR scratch spills in the
// decision-safe kernel, -25 % throughput) instead of a scalar-cache hit.
#ifndef ZOIC_GUARD_PIN
#define ZOIC_GUARD_PIN 1
#endif
#ifndef ZOIC_TRACE_PREFETCH
#define ZOIC_TRACE_PREFETCH 1
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
    S.center = t[i].center; S.radius2 = t[i].radius2; S.sign = t[i].sign; S.housing2 = t[i].housing2;
    S.eta = t[i].eta; S.qOffset = t[i].qOffset; S.krScale = t[i].krScale;
    S.housingLo = t[i].housingLo; S.housingHi = t[i].housingHi;
    S.pad0 = S.pad1 = S.pad2 = 0.0f;
    return S;
}

// every word of S is in its SGPR from here on (the compiler puts the s_waitcnt of the scalar loads in front of this)
__device__ __forceinline__ void surface_arrived(const FastSurface &S)
{
    asm volatile("" : : "s"(S.center), "s"(S.radius2), "s"(S.sign), "s"(S.housing2), "s"(S.eta), "s"(S.qOffset), "s"(S.krScale),
                 "s"(S.housingLo), "s"(S.housingHi));
}

__device__ __forceinline__ float uniform_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ FastSurface uniform_surface(const FastSurface &s)
{
    FastSurface r;
    r.center = uniform_f32(s.center); r.radius2 = uniform_f32(s.radius2); r.sign = uniform_f32(s.sign);
    r.housing2 = uniform_f32(s.housing2); r.eta = uniform_f32(s.eta); r.qOffset = uniform_f32(s.qOffset);
    r.krScale = uniform_f32(s.krScale); r.housingLo = uniform_f32(s.housingLo); r.housingHi = uniform_f32(s.housingHi);
    r.pad0 = r.pad1 = r.pad2 = 0.0f;
    return r;
}

// One ray through the lens, rear -> front (traceThroughLensElements, zoic.cpp:1099-1158).
//   o : ray origin, advanced to the last accepted hit point (zoic.cpp:1130)
//   d : raw direction on entry; the refracted unit direction on exit.  If the ray dies before its first refraction d is
//       left untouched -- the partial state the reference hands out for rays that run out of tries (zoic.cpp:1951-1961).
// With |u| = 1 and the hit on the sphere:  N = (c - hit)/R,  cos(i) = -(u.N) = thc/|R|  (no dot product),
//   1 - cs2 = (1 - eta^2) + (eta/R)^2 thc^2 = (eta/R)^2 q,   q = thc^2 + (1 - eta^2) R^2 / eta^2,   TIR <=> q < 0,
//   u' = eta u + kr (c - hit),   kr = (eta cos(i) - sqrt(1 - cs2)) / R = eta/(|R| R) (thc - sqrt(q))      (|u'| = 1 again).
// The squared distance of the origin from the axis is carried from interface to interface (it is the h^2 of the previous
// hit): |L|^2 = h^2 + Lz^2.  Per interface: 26 VALU + 2 v_sqrt_f32 incl. 2 compares (3 with a guard band): a sphere miss needs none of its own (fast_hit).
//
// FastHit: the arithmetic of ONE interface, shared by the predicated trace, the branchy trace and the interface-0 test, so that
// the three agree bit for bit on every decision.
__device__ __forceinline__ float fast_norm2(const V3 &d) { return ffma(d.z, d.z, ffma(d.y, d.y, d.x * d.x)); }
__device__ __forceinline__ float fast_axis2(const V3 &o) { return ffma(o.y, o.y, o.x * o.x); }
// a near-planar interface (the stop): its `sign` word is 2R instead of +-1 (lens_system.cpp fill_surfaces)
__device__ __forceinline__ bool surface_is_flat(const FastSurface &S) { return (__builtin_bit_cast(uint32_t, S.sign) & 0x7fffffffu) != 0x3f800000u; }
struct FastHit { float w, thc, h2, tca; V3 hit; };
__device__ __forceinline__ FastHit fast_hit(const FastSurface &S, const V3 &o, float oAxis2, const V3 &u)
{
    FastHit r;
    const float Lz = S.center - o.z;
    float tca, t;
#if ZOIC_FAST_STABLE_STOP
    // EXPERIMENT (tables.hpp ZOIC_FAST_STABLE_STOP, off in the product build; DESIGN section 6, profiles/ab_r05/ab_stop.log): another root at the
    // stop (zoic.cpp:933: radius 0 -> a sphere of |R| ~ 1e4 cm).  t = tca + sgn(R) thc subtracts two numbers of magnitude |R| and leaves
    // the hit good to ulp(|R|) ~ 1e-3 cm: the clip there is decided by the REFERENCE's rounding noise, which is why the interface
    // carries a guard band (lens_system.cpp) as wide as FAST's and the reference's hits can differ.  Neither variant narrows it.
    if (__builtin_expect(surface_is_flat(S), 0)) {          // wave-uniform (S is in SGPRs): a scalar compare and a branch not taken
        const float twoR = S.sign;                          // flat surfaces carry 2R where the others carry sgn(R) (tables.hpp)
        const float sgn = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, twoR) & 0x80000000u) | 0x3f800000u);
#if ZOIC_FAST_STABLE_STOP == 1
        // Variant 1: the root in its conjugate form, t = C / (tca - sgn(R) thc), C = h0^2 + g (g - 2R), g = (c + R) - o.z -- good to ~2
        // ulps of t.  FAST/STRICT disagreements at the stop went UP (C4: 5349 -> 9958 of 16.8 M rays): the plain root makes nearly the
        // same rounding errors as the reference's (same cancellation on the same operands) and they cancel in the comparison; an
        // accurate hit differs from the reference's by the reference's whole noise.
        tca = ffma(-o.y, u.y, ffma(Lz, u.z, -(o.x * u.x)));
        const float d2 = ffma(-tca, tca, ffma(Lz, Lz, oAxis2));
        r.w = S.radius2 - d2;
        r.thc = fsqrt_fast(r.w);
        const float R = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, twoR) - 0x00800000u);
        const float g = (S.center + R) - o.z;
        const float C = ffma(g, g - twoR, oAxis2);
        const float plain = ffma(r.thc, sgn, tca), den = ffma(-r.thc, sgn, tca);
        t = (tca * sgn < 0.0f) ? C * frcp_fast(den) : plain;
#else
        // Variant 2: the reference's OWN operations in its order, one rounding per operator, a correctly rounded root (zoic.cpp:975-991),
        // on FAST's (o, u).  Disagreements DOWN (C2 582 -> 250, C5 92 -> 40, C4 5349 -> 4433) but their largest margin is unchanged
        // (one quantum of ulp(|R|): the two arrive with directions a last bit apart, and Lz u.z then rounds one ulp apart), so the
        // band -- sized by the largest -- and the work list stay as they are, and the extra instructions cost C4 / C5 1-2.5 %.
        const float Lx = 0.0f - o.x, Ly = 0.0f - o.y;
        tca = (Lx * u.x + Ly * u.y) + Lz * u.z;
        const float d2 = ((Lx * Lx + Ly * Ly) + Lz * Lz) - (tca * tca);
        r.w = S.radius2 - d2;
        r.thc = ZOIC_SQRT_RN(r.w);                          // (a miss -- w < 0 -- stays a NaN hit: see below)
        t = tca + r.thc * sgn;
#endif
    } else
#endif
    {
        tca = ffma(-o.y, u.y, ffma(Lz, u.z, -(o.x * u.x)));     // L.u with L = (-o.x, -o.y, Lz)
        const float d2 = ffma(-tca, tca, ffma(Lz, Lz, oAxis2)); // |L|^2 - tca^2, |L|^2 = h^2 of the previous hit + Lz^2
        // A ray that MISSES the sphere (d2 > radius2, zoic.cpp:981) takes the root of a negative number: thc, the hit point and h^2
        // are NaN, and every clip test below is written !(h2 <= limit) -- true for NaN -- so the miss needs no compare of its own.
        // (A ray that ARRIVES as NaN -- a lens sample at the disk mapping's 0/0 centre, zoic.cpp:697-699 -- passes every `>` of the
        // reference and comes out a NaN success; it is recognised by its NaN tca at the first interface, is_nan_ray.)
        r.w = S.radius2 - d2;                               // thc^2
        r.thc = fsqrt_fast(r.w);
        t = ffma(r.thc, S.sign, tca);
    }
    r.tca = tca;
    r.hit = V3{ffma(u.x, t, o.x), ffma(u.y, t, o.y), ffma(u.z, t, o.z)};
    r.h2 = ffma(r.hit.y, r.hit.y, r.hit.x * r.hit.x);
    return r;
}
__device__ __forceinline__ bool is_nan_ray(const FastHit &h) { return h.tca != h.tca; }

// Snell at the hit point: returns q (TIR <=> q < 0) and writes the refracted unit direction
__device__ __forceinline__ float fast_refract(const FastSurface &S, const FastHit &h, V3 &u)
{
    const float q = h.w + S.qOffset;
    const float kr = (h.thc - fsqrt_fast(fabsf(q))) * S.krScale;
    u = V3{ffma(u.x, S.eta, -(h.hit.x * kr)), ffma(u.y, S.eta, -(h.hit.y * kr)), ffma(S.center - h.hit.z, kr, u.z * S.eta)};
    return q;
}

// Decision-safe mode: only ill-conditioned interfaces carry a guard band (housingLo < housingHi, set by the host: in practice
// the stop, lens_system.cpp fill_surfaces).  h^2 in (housingLo, housingHi] is too close to call; on unguarded interfaces both
// edges equal housing2 and the interval is empty.
__device__ __forceinline__ bool surface_is_guarded(const FastSurface &S) { return S.housingLo < S.housingHi; }

// One interface of the branchy trace.  Returns 0 = passed, 1 = clipped (o, u untouched), 2 = total internal reflection
// (o advanced, u untouched) -- the two partial states the reference can leave.
__device__ __forceinline__ int fast_interface(const FastSurface &S, V3 &o, float &oAxis2, V3 &u, bool *near = nullptr)
{
    const FastHit h = fast_hit(S, o, oAxis2, u);
    if (near) *near = !(h.h2 <= S.housingLo) & (h.h2 <= S.housingHi);
    if (!(h.h2 <= S.housing2) && !is_nan_ray(h)) return 1;   // sphere miss (NaN h2) or housing clip; the stop's housing2 includes the user aperture
    o = h.hit;
    oAxis2 = h.h2;
    V3 un = u;
    if (fast_refract(S, h, un) < 0.0f) return 2;             // cs2 > 1 (only reachable when eta > 1)
    u = un;
    return 0;
}

// Does a ray clear interface 0 (first sphere hit + rear-element housing)?  The first half of an interface; used by the
// kernel's candidate search so that tries dying at the rear element never pay for a trace.  GUARD: `near` is set when the
// housing decision lies inside its guard band (the answer is then not used: the ray goes to STRICT).
template <bool GUARD>
__device__ __forceinline__ bool interface0_clear_fast(const FastSurface &S, V3 o, V3 d, bool &near)
{
    const float inv = frsq_fast(fast_norm2(d));
    const V3 u{d.x * inv, d.y * inv, d.z * inv};
    const FastHit h = fast_hit(S, o, fast_axis2(o), u);
    if constexpr (GUARD) {
        const bool insideLo = h.h2 <= S.housingLo;           // false for a sphere miss (NaN)
        near = !insideLo & (h.h2 <= S.housingHi);
        return insideLo | is_nan_ray(h);
    } else {
        near = false;
        return (h.h2 <= S.housing2) | is_nan_ray(h);
    }
}

// Rolled, branchy trace for any interface count.  It leaves exactly the partial state of the reference on every exit
// path, so it is also what finishes rays that ran out of tries in the predicated kernel below.
__device__ __forceinline__ bool trace_lens_fast_rolled(const KolbTable &T, V3 &o, V3 &d, uint32_t &tirCount, bool *unsure = nullptr)
{
    const float inv = frsq_fast(fast_norm2(d));
    V3 u{d.x * inv, d.y * inv, d.z * inv};
    float oAxis2 = fast_axis2(o);
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
        const int r = fast_interface(S, o, oAxis2, u, unsure ? &near : nullptr);
        if (unsure) *unsure |= near;
        if (r != 0) { if (r == 2) ++tirCount; ok = false; break; }
        refracted = true;
        if (++i == n) break;
    }
    if (refracted) d = u;
    return ok;
}

// The same from interface `first` on (the listed kernel continues behind the stop, kolb_listed_body.hpp): d is normalised like a
// fresh direction; the table words come through the kernarg pointer (a run-time index into the by-value KolbTable makes the
// compiler copy its arrays to scratch: 2.9 KB per lane and a listed kernel six times slower, round 4).
__device__ __forceinline__ bool trace_lens_fast_rolled_from(FastSurfaceTable surf, int n, int first, V3 &o, V3 &d, uint32_t &tirCount, bool *unsure)
{
    const float inv = frsq_fast(fast_norm2(d));
    V3 u{d.x * inv, d.y * inv, d.z * inv};
    float oAxis2 = fast_axis2(o);
    bool ok = true, refracted = false;
    for (int i = first; i < n;) {
        const int iu = __builtin_amdgcn_readfirstlane(i);
        const FastSurface S = load_surface<false>(surf, iu);
        bool near = false;
        const int r = fast_interface(S, o, oAxis2, u, &near);
        *unsure |= near;
        if (r != 0) { if (r == 2) ++tirCount; ok = false; break; }
        refracted = true;
        ++i;
    }
    if (refracted) d = u;
    return ok;
}

// Predicated, fully unrolled trace for a lens with exactly NS interfaces: NO divergent control flow.  Every lane
// evaluates every interface; a lane that is clipped or totally reflected only clears its bit in the `alive` mask.  The
// masks are plain 64-bit scalars (one v_cmp into an SGPR pair per decision, s_and / s_andn2 / s_or to combine them): no
// v_cndmask, no exec-mask bookkeeping.  Dead lanes keep computing on garbage, which costs nothing: a wave issues each VALU
// instruction once whatever its exec mask.  The table words are s_loads at fixed kernel-argument offsets (SGPR operands:
// staging the table in LDS instead was measured 36 % slower on C4 -- VGPR copies, no scalar operands).  A wave-uniform
// test every second interface leaves the trace as soon as no lane is alive (heavily vignetted passes).
// Returns the alive mask; o/d are the exit point and unit direction for alive lanes (unspecified for dead ones -- rays that
// finish dead get their reference partial state from trace_lens_fast_rolled).  tirMask: lanes that were totally reflected.
// GUARD (decision-safe mode): unsureMask collects the lanes still alive at a guarded interface whose clip decision lies inside
// the guard band.
template <int NS, bool GUARD = false>
__device__ __forceinline__ unsigned long long trace_lens_fast_pred(FastSurfaceTable surf, V3 &o, V3 &d, unsigned long long alive0,
                                                                   unsigned long long &tirMask, unsigned long long &unsureMask)
{
    static_assert(NS > 0, "predicated trace needs a compile-time interface count");
    const float inv = frsq_fast(fast_norm2(d));
    V3 u{d.x * inv, d.y * inv, d.z * inv};
    float oAxis2 = fast_axis2(o);
    unsigned long long alive = alive0, tirSeen = 0ull, unsure = 0ull;   // alive0: lanes without a candidate ride along dead
    unsigned long long nanRays = 0ull;                                  // candidates that arrived as NaN: they "pass" everything
    // The table words of interface i + 1 are requested BEFORE interface i is evaluated (scalar loads return out of order, so
    // the only wait there is is lgkmcnt(0): `surface_arrived` takes it at the end of interface i, a whole interface -- ~30 VALU --
    // after the request; the sched_barrier keeps the scheduler from sinking the request towards its use).  Loading at the use
    // instead stalls every wave for a scalar-cache round trip per interface.
    FastSurface S = load_surface<true>(surf, 0);
    if constexpr (ZOIC_TRACE_PREFETCH != 0) surface_arrived(S);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (i >= 2 && (i & 1) == 0 && alive == 0ull) break;   // wave-uniform early out, every 2nd interface
        FastSurface Sn = S;
        if constexpr (ZOIC_TRACE_PREFETCH != 0) {
            if (i + 1 < NS) { Sn = load_surface<true>(surf, i + 1); __builtin_amdgcn_sched_barrier(0); }   // the request goes out HERE
        }
        const FastHit h = fast_hit(S, o, oAxis2, u);
        if (i == 0) nanRays = __ballot(is_nan_ray(h));
        unsigned long long clipped;   // sphere miss or housing clip: ONE compare (a miss makes h2 NaN, fast_hit)
        if constexpr (GUARD) {
            // two compares: not inside the band's lower edge = clipped or too close to call; inside its upper edge as well =
            // too close to call (then the ray goes to STRICT and `clipped` is never used).  Unguarded: both edges = housing2.
            clipped = ~__ballot(h.h2 <= S.housingLo);
            unsure |= alive & clipped & __ballot(h.h2 <= S.housingHi);
        } else clipped = ~__ballot(h.h2 <= S.housing2);   // the stop's housing2 includes the user aperture
        o = h.hit;
        oAxis2 = h.h2;
        const unsigned long long tirHere = __ballot(fast_refract(S, h, u) < 0.0f);
        alive &= ~clipped;
        tirSeen |= alive & tirHere;                      // counted only by rays that reached the refraction
        alive &= ~tirHere;
        if constexpr (ZOIC_TRACE_PREFETCH != 0) { if (i + 1 < NS) surface_arrived(Sn); S = Sn; }   // ... and is waited for here, in the same block
        else if (i + 1 < NS) S = load_surface<GUARD && (ZOIC_GUARD_PIN != 0)>(surf, i + 1);
    }
    tirMask = tirSeen;
    unsureMask = unsure;
    d = u;
    return alive | (alive0 & nanRays);
}

// The same trace for R rays PER LANE, keeping a failed ray's partial state (the resident tile workers, mailbox.hip).
//   * R rays per lane: a resident wave is alone on its SIMD and retires a dependent instruction every ~10 cycles, so a second,
//     independent ray in the same lanes rides in the first one's latency (two tries of a ray evaluated in the time of ~1.2);
//   * KEEP: a lane that dies keeps the PARTIAL state the branchy trace leaves on the same exit -- clipped: (o, u) as they arrived at that
//     interface; totally reflected: o advanced to the hit, u as it arrived; d = the raw direction if the ray never got through a
//     refraction, the last refracted unit direction otherwise -- so that a ray that finishes failed (try 26, zoic.cpp:1951-1961) needs
//     no second, branchy trace: six v_cndmask per interface off the chain against a whole round on it.
// The arithmetic per interface is the same FastHit / fast_refract as everywhere: same bits.  alive[r]: in = the lanes whose ray r is a
// candidate, out = the lanes whose ray r got through (a ray that arrived NaN is clipped nowhere, as in the branchy trace: it stays
// alive and its state goes NaN).
template <int NS, bool GUARD, int R>
__device__ __forceinline__ void trace_lens_fast_pred_keep(FastSurfaceTable surf, V3 (&o)[R], V3 (&d)[R], unsigned long long (&alive)[R],
                                                          unsigned long long (&tirMask)[R], unsigned long long (&unsureMask)[R])
{
    static_assert(NS > 0 && R >= 1, "predicated trace needs a compile-time interface count");
    V3 u[R];
    float oAxis2[R];
    unsigned long long nanRays[R], refracted[R];
    unsigned long long any = 0ull;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float inv = frsq_fast(fast_norm2(d[r]));
        u[r] = V3{d[r].x * inv, d[r].y * inv, d[r].z * inv};
        oAxis2[r] = fast_axis2(o[r]);
        tirMask[r] = 0ull; unsureMask[r] = 0ull; nanRays[r] = 0ull; refracted[r] = 0ull;
        any |= alive[r];
    }
    FastSurface S = load_surface<true>(surf, 0);
    surface_arrived(S);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (i >= 2 && (i & 1) == 0 && any == 0ull) break;   // wave-uniform early out, every 2nd interface
        FastSurface Sn = S;
        if (i + 1 < NS) { Sn = load_surface<true>(surf, i + 1); __builtin_amdgcn_sched_barrier(0); }   // the next interface's words are requested HERE
        any = 0ull;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const FastHit h = fast_hit(S, o[r], oAxis2[r], u[r]);
            if (i == 0) nanRays[r] = __ballot(is_nan_ray(h));
            unsigned long long clipped;
            if constexpr (GUARD) {
                clipped = ~__ballot(h.h2 <= S.housingLo);
                unsureMask[r] |= alive[r] & clipped & __ballot(h.h2 <= S.housingHi);
            } else clipped = ~__ballot(h.h2 <= S.housing2);
            alive[r] &= ~(clipped & ~nanRays[r]);
            const bool through = __builtin_amdgcn_inverse_ballot_w64(alive[r]);
            o[r] = V3{through ? h.hit.x : o[r].x, through ? h.hit.y : o[r].y, through ? h.hit.z : o[r].z};
            oAxis2[r] = h.h2;
            V3 un = u[r];
            const unsigned long long tirHere = __ballot(fast_refract(S, h, un) < 0.0f);
            tirMask[r] |= alive[r] & tirHere;                // counted only by rays that reached the refraction
            alive[r] &= ~tirHere;
            const bool bent = __builtin_amdgcn_inverse_ballot_w64(alive[r]);
            u[r] = V3{bent ? un.x : u[r].x, bent ? un.y : u[r].y, bent ? un.z : u[r].z};
            if (i == 0) refracted[r] = alive[r];
            any |= alive[r];
        }
        if (i + 1 < NS) surface_arrived(Sn);
        S = Sn;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { if (__builtin_amdgcn_inverse_ballot_w64(refracted[r])) d[r] = u[r]; }
}

}  // namespace zoic

#pragma clang fp contract(off)
