// kolb_device.hpp -- device helpers shared by the persistent Kolb kernels (kolb_refill.hip, kolb_queue.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "device_search.hpp"
#include "exact_math.hpp"
#include "fast_optics.hpp"
#include "kernels.hpp"
#include "optics.hpp"
#include "ray_store.hpp"

#pragma STDC FP_CONTRACT OFF

namespace zoic {

#ifndef ZOIC_REFILL_BLOCK
#define ZOIC_REFILL_BLOCK 256   // experiments: -DZOIC_REFILL_BLOCK=128 / 512 (with -DZOIC_GRID_BLOCKS=4096 / 1024 for the same number of waves)
#endif
constexpr int kRefillBlock = ZOIC_REFILL_BLOCK;
constexpr int kWavesPerBlock = kRefillBlock / 64;
constexpr uint32_t kLutLdsWords = 2 * kLutEntries;  // (maxScale, centroid.x) pairs at the start of the dynamic LDS
#ifndef ZOIC_MIN_SEARCHING
#define ZOIC_MIN_SEARCHING 16
#endif
constexpr uint32_t kMinSearching = ZOIC_MIN_SEARCHING;  // the candidate search goes on while at least this many lanes of the wave are looking

template <bool STRICT>
__device__ __forceinline__ V2 lens_sample(const KolbTable &T, const BokehTables &B, const float *bokehLds, float u, float v)
{
    if (T.useImage) {
        if (bokehLds) return bokeh_sample_cells<STRICT>(B, bokehLds, T.bokehW, T.bokehH, u, v);
        return bokeh_sample_device(B, T.bokehW, T.bokehH, u, v);
    }
    if constexpr (STRICT) return concentric_disk(u, v);
    else return concentric_disk_f32(u, v);
}

extern __shared__ __align__(16) float zoicDynLds[];

// lut_lookup (optics.hpp, zoic.cpp:1891-1911) reading the (maxScale, centroid.x) pairs from LDS: identical arithmetic
__device__ __forceinline__ bool lut_lookup_lds(const float2 *lut, int lutSize, float dist, float &maxScale, float &translation)
{
    const float samplingErrorCorrection = 1.05f;
    const float scaled = dist * 8.0f;
    const int low = static_cast<int>(ceilf(scaled));
    if (!(scaled <= static_cast<float>(lutSize - 1))) { maxScale = 0.0f; translation = 0.0f; return false; }
    if (low <= 0) {
        const float2 e = lut[0];
        maxScale = e.x * samplingErrorCorrection; translation = e.y;
        return true;
    }
    const float4 pr = *reinterpret_cast<const float4 *>(lut + (low - 1));   // entries low-1 (xy) and low (zw)
    const float percentage = (dist - static_cast<float>(low) * 0.125f) * -8.0f;
    maxScale = (pr.z + percentage * (pr.x - pr.z)) * samplingErrorCorrection;
    translation = pr.w + percentage * (pr.y - pr.w);
    return true;
}

// STRICT arithmetic (optics.hpp trace_lens_strict, operation for operation) in the predicated, fully unrolled shape of
// trace_lens_fast_pred: lanes that fail only clear their bit in `alive`; every surviving lane executes exactly the
// reference's sequence of roundings, so alive lanes are bit-identical to the branchy version.  Rays that FINISH failed
// get their partial state from trace_lens_strict.
//
// The four square roots and three reciprocals per interface use the LEAN correctly-rounded sequences of exact_math.hpp
// without their per-call range guard (a guard branch would be taken by every wave: dead lanes ride along on garbage).
// Instead one integer compare per root records whether an ALIVE lane ever left the verified range [1e-30, 1e30]
// (`outOfRange`); the caller then re-runs that try through trace_lens_strict, whose roots are guarded.  In range the lean
// sequences ARE the IEEE results (tools/ubench/exact_math_check.hip, all 2^32 inputs), so the bits do not change.
__device__ __forceinline__ bool lean_in_range(float x)   // positive floats order like their bit patterns; NaN/inf/0/denormals fall outside
{
    return (__builtin_bit_cast(uint32_t, x) - __builtin_bit_cast(uint32_t, kExactLo)) <=
           (__builtin_bit_cast(uint32_t, kExactHi) - __builtin_bit_cast(uint32_t, kExactLo));
}
__device__ __forceinline__ V3 normalize3_lean(V3 a, bool &inRange)
{
    const float s = a.x * a.x + a.y * a.y + a.z * a.z;
    inRange = lean_in_range(s);                 // then sqrt(s) lies in [1e-15, 1e15], inside the reciprocal's range
    const float t = rcp_rn_lean(sqrt_rn_lean(s));
    return V3{a.x * t, a.y * t, a.z * t};
}

// the same for a vector that is unit to rounding already (exact_math.hpp rcp_sqrt_rn_near_one): the same bits, no root at all
__device__ __forceinline__ V3 normalize3_unit(V3 a, bool &inRange)
{
    const float s = a.x * a.x + a.y * a.y + a.z * a.z;
    const float t = rcp_sqrt_rn_near_one(s, inRange);
    return V3{a.x * t, a.y * t, a.z * t};
}

// interface0_clear_strict (optics.hpp) with the lean roots; `inRange` false: the caller repeats the test with the guarded ones
__device__ __forceinline__ bool interface0_clear_strict_lean(const KolbTable &T, V3 o, V3 d, bool &inRange)
{
    const Surface S = T.surf[0];
    bool r0;
    V3 u = normalize3_lean(d, r0);
    V3 L{0.0f - o.x, 0.0f - o.y, S.center - o.z};
    float tca = dot3(L, u);
    float d2 = dot3(L, L) - (tca * tca);
    const float w = fabsf(S.radius2 - d2);
    inRange = r0 & lean_in_range(w);
    float thc = sqrt_rn_lean(w);
    float t = tca + thc * S.sign;
    V3 hit{o.x + u.x * t, o.y + u.y * t, o.z + u.z * t};
    float h2 = hit.x * hit.x + hit.y * hit.y;
    return !((d2 > S.radius2) | (h2 > S.housing2));
}

// RANGE (the listed kernel, kolb_listed_body.hpp): only interfaces first ... last (wave-uniform run-time bounds) are traced
template <int NS, bool RANGE = false>
__device__ __forceinline__ bool trace_lens_strict_pred(const KolbTable &T, V3 &o, V3 &d, uint32_t &tirCount, bool alive0, bool &outOfRange,
                                                       int first = 0, int last = NS - 1)
{
    static_assert(NS > 0, "predicated trace needs a compile-time interface count");
    bool alive = alive0, tirSeen = false, anyAlive = true, oor = false;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if constexpr (RANGE) { if (i < first || i > last) continue; }
        if (i >= 2 && (i & 1) == 0) anyAlive = __ballot(alive) != 0ull;
        if (!anyAlive) continue;
        const Surface &S = T.surf[i];
        bool r0, r1, r2, r3;
        // behind the first interface d is Snell's output: unit to rounding (a lane where it is not -- never seen -- sets oor and the
        // caller repeats the try through the guarded branchy trace)
        V3 u = (i == 0) ? normalize3_lean(d, r0) : normalize3_unit(d, r0);
        V3 L{0.0f - o.x, 0.0f - o.y, S.center - o.z};
        float tca = dot3(L, u);
        float d2 = dot3(L, L) - (tca * tca);
        const float w = fabsf(S.radius2 - d2);
        r1 = lean_in_range(w);
        float thc = sqrt_rn_lean(w);
        float t = tca + thc * S.sign;
        V3 hit{o.x + u.x * t, o.y + u.y * t, o.z + u.z * t};
        float h2 = hit.x * hit.x + hit.y * hit.y;
        const bool clipped = (d2 > S.radius2) | (h2 > S.housing2);   // the stop's housing2 includes the user aperture
        V3 nrm = normalize3_lean(V3{0.0f - hit.x, 0.0f - hit.y, S.center - hit.z}, r2);
        nrm = V3{nrm.x * S.sign, nrm.y * S.sign, nrm.z * S.sign};
        o = hit;
        V3 N = normalize3_unit(nrm, r3);      // zoic.cpp:1010: the normal, normalised a second time
        oor |= alive & !(r0 & r1 & r2 & r3);   // a lane still alive HERE consumed these roots
        float c1 = -dot3(u, N);
        // cs2 = (float)((double)(eta*eta) * (1.0 - (double)(c1*c1))), zoic.cpp:1016.  For c1^2 >= 1/32 the f64 difference has at most
        // 29 significant bits and its f64 product with the 24-bit eta^2 is EXACT, so the conversion to float is the only rounding
        // of eta^2 - eta^2 c1^2: one fmaf gives the same bits (no converts, no f64).  Grazing incidence (c1^2 < 1/32: the product
        // may round in f64 first) takes the reference's expression, wave-uniformly.
        const float c1sq = c1 * c1, eta2 = S.eta * S.eta;
        float cs2 = __builtin_fmaf(-eta2, c1sq, eta2);
        const bool grazing = !(c1sq >= 0.03125f);
        if (__builtin_expect(__ballot(alive && grazing) != 0ull, 0)) {
            const float cs2d = static_cast<float>(static_cast<double>(eta2) * (1.0 - static_cast<double>(c1sq)));
            cs2 = grazing ? cs2d : cs2;
        }
        const bool tirHere = (S.tirPossible != 0u) & (cs2 > 1.0f);
        tirSeen |= alive & !clipped & tirHere;
        alive &= !clipped & !tirHere;
        // f64 sqrt of |1 - cs2| through the lean sequence (verified on every float cs2 with |1 - cs2| == 0 or in [1e-30, 1e30])
        oor |= alive & !(fabsf(cs2) <= 1.0e30f);
        float k = static_cast<float>(static_cast<double>(S.eta * c1) - sqrt64_rn_lean(fabs(1.0 - static_cast<double>(cs2))));
        d = V3{u.x * S.eta + N.x * k, u.y * S.eta + N.y * k, u.z * S.eta + N.z * k};
    }
    tirCount += tirSeen ? 1u : 0u;
    outOfRange = oor;
    return alive;
}

}  // namespace zoic
