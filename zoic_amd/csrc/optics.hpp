// optics.hpp -- STRICT-mode lens arithmetic shared by the host precompute (exit-pupil LUT build) and the
// strict HIP kernels.  "Strict" = the reference's operation order, one IEEE rounding per written operator,
// its f64 intermediates where zoic.cpp has them, no FMA contraction (the library is compiled with
// -ffp-contract=off and this header repeats it).  Results are bit-identical on x86-64 and gfx950.
//
// Arnold SDK inlines used by the reference (AiV3Dot / AiV3Normalize / AtVector operators) are restated as
// dot3 / normalize3 below; see DESIGN.md "third-party arithmetic".
#pragma once
#include <cmath>
#include <cstdint>

#include "tables.hpp"

// Correctly rounded f32 sqrt and reciprocal of the STRICT arithmetic: on the device the lean sequences of exact_math.hpp
// (exhaustively verified equal to IEEE sqrt / divide, guarded outside their range), on the host the C library / compiler.
#if defined(__HIP_DEVICE_COMPILE__)
#include "exact_math.hpp"
#define ZOIC_SQRT_RN(x) ::zoic::sqrt_rn(x)
#define ZOIC_RCP_RN(x) ::zoic::rcp_rn(x)
#else
#define ZOIC_SQRT_RN(x) sqrtf(x)
#define ZOIC_RCP_RN(x) (1.0f / (x))
#endif

#pragma STDC FP_CONTRACT OFF

namespace zoic {

struct V3 { float x, y, z; };
struct V2 { float x, y; };

constexpr float kPi = 3.14159265358979323846f;       // AI_PI
constexpr float kTwoPi = kPi * 2;                     // AI_PI * 2 (f32 product, zoic.cpp:662)
constexpr float kPiOver2 = 1.57079632679489661923f;   // AI_PIOVER2
constexpr float kInv2p32 = 2.3283064365386963e-10f;   // 1/4294967296, exact power of two

ZOIC_HD float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

ZOIC_HD V3 normalize3(V3 a)
{
    float t = ZOIC_SQRT_RN(a.x * a.x + a.y * a.y + a.z * a.z);
    if (t != 0.0f) t = ZOIC_RCP_RN(t);
    return V3{a.x * t, a.y * t, a.z * t};
}

// ---- xorshift128, zoic.cpp:647-652 -------------------------------------------------------------
struct Rng { uint32_t x, y, z, w; };

ZOIC_HD void rng_seed_reference(Rng &r) { r = Rng{123456789u, 362436069u, 521288629u, 88675123u}; }

ZOIC_HD uint32_t xor128(Rng &r)
{
    uint32_t t = r.x ^ (r.x << 11);
    r.x = r.y; r.y = r.z; r.z = r.w;
    r.w = r.w ^ (r.w >> 19) ^ t ^ (t >> 8);
    return r.w;
}

// xor128()/4294967296.0 narrowed to the float argument (zoic.cpp:1881,1930) == (float)u * 2^-32 == the
// f32 divide of zoic.cpp:1806/1411: u -> f32 is the only rounding, the scale is exact.  Can return 1.0f.
ZOIC_HD float rng_unit(uint32_t u) { return static_cast<float>(u) * kInv2p32; }

// 32-bit PCG output hash: seeds the per-ray retry streams and the synthetic samples
ZOIC_HD uint32_t pcg_hash(uint32_t v)
{
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

// private retry stream of ray `gid`: never all-zero
ZOIC_HD Rng rng_for_ray(uint32_t seed, uint64_t gid)
{
    uint32_t lo = static_cast<uint32_t>(gid), hi = static_cast<uint32_t>(gid >> 32);
    uint32_t k = pcg_hash(seed ^ pcg_hash(hi + 0x9E3779B9u));
    Rng r;
    r.x = pcg_hash(k ^ (lo * 4u + 0u));
    r.y = pcg_hash(k ^ (lo * 4u + 1u) ^ 0x85EBCA6Bu);
    r.z = pcg_hash(k ^ (lo * 4u + 2u) ^ 0xC2B2AE35u);
    r.w = pcg_hash(k ^ (lo * 4u + 3u) ^ 0x27D4EB2Fu) | 1u;
    return r;
}

// ---- fastSin / fastCos, zoic.cpp:661-681 --------------------------------------------------------
// fmod(x + AI_PI, AI_PI*2) for x + AI_PI in [0, 4pi): the subtraction is exact (Sterbenz), so this equals
// the C fmod the reference calls for every argument the hot path produces (phi in [-pi/4, 3pi/4],
// theta in [-pi, pi], +pi/2 for the cosine); NaN propagates.
ZOIC_HD float wrap_to_pi(float x)
{
    float v = x + kPi;
    if (v >= kTwoPi) v = v - kTwoPi;
    return v - kPi;
}

ZOIC_HD float parabola_sin(float x)  // body of fastSin after range reduction, zoic.cpp:663-667
{
    const float B = 4.0f / kPi;
    const float C = -4.0f / (kPi * kPi);
    float y = B * x + C * x * fabsf(x);
    const float P = 0.225f;
    return P * (y * fabsf(y) - y) + y;
}

ZOIC_HD float fast_sin(float x) { return parabola_sin(wrap_to_pi(x)); }

ZOIC_HD float fast_cos(float x)
{
    x = static_cast<float>(static_cast<double>(x) + static_cast<double>(kPi) * 0.5);  // zoic.cpp:673
    return parabola_sin(wrap_to_pi(x));
}

// concentricDiskSample, zoic.cpp:686-704
ZOIC_HD V2 concentric_disk(float ox, float oy)
{
    float a = static_cast<float>(2.0 * static_cast<double>(ox) - 1.0);
    float b = static_cast<float>(2.0 * static_cast<double>(oy) - 1.0);
    float r, phi;
    if ((a * a) > (b * b)) {
        r = a;
        phi = 0.78539816339f * (b / a);
    } else {
        r = b;
        phi = kPiOver2 - 0.78539816339f * (a / b);
    }
    return V2{r * fast_cos(phi), r * fast_sin(phi)};
}

// ---- traceThroughLensElements, zoic.cpp:1099-1158 (== ...ForApertureSize, zoic.cpp:1309-1350) ------
// o/d are updated in place exactly as the reference leaves them on every exit path (the caller relies on
// the partial state when a ray exhausts its tries, zoic.cpp:1951-1961).
// (first, last: the interfaces to trace, inclusive -- the whole lens for the reference's loop; the listed kernel of the
// decision-safe FAST mode traces up to the stop and from the stop on, kolb_listed_body.hpp)
ZOIC_HD bool trace_lens_strict_range(const KolbTable &T, V3 &o, V3 &d, uint32_t &tirCount, int first, int last)
{
    for (int ii = first; ii <= last; ++ii) {
#if defined(__HIP_DEVICE_COMPILE__)
        // all lanes still looping are at the same surface: keep the index (and the table fetch) scalar
        const int i = __builtin_amdgcn_readfirstlane(ii);
#else
        const int i = ii;
#endif
        const Surface S = T.surf[i];
        // raySphereIntersection(.., reverse=false, tracingRealRays=true), zoic.cpp:973-995
        V3 u = normalize3(d);
        V3 L{0.0f - o.x, 0.0f - o.y, S.center - o.z};
        float tca = dot3(L, u);
        float d2 = dot3(L, L) - (tca * tca);
        if (d2 > S.radius2) return false;
        float thc = ZOIC_SQRT_RN(fabsf(S.radius2 - d2));
        float t = tca + thc * S.sign;
        V3 hit{o.x + u.x * t, o.y + u.y * t, o.z + u.z * t};
        // housing / user aperture clip, zoic.cpp:1111-1117
        float h2 = hit.x * hit.x + hit.y * hit.y;
        if (h2 > S.housing2) return false;  // housing2 of the stop already holds min(housing, user aperture)^2
        // intersectionNormal, zoic.cpp:999-1004
        V3 nrm = normalize3(V3{0.0f - hit.x, 0.0f - hit.y, S.center - hit.z});
        nrm = V3{nrm.x * S.sign, nrm.y * S.sign, nrm.z * S.sign};
        o = hit;  // zoic.cpp:1130
        // calculateTransmissionVector, zoic.cpp:1008-1025 (its normalise of the incident vector == u)
        V3 N = normalize3(nrm);
        float c1 = -dot3(u, N);
        float cs2 = static_cast<float>(static_cast<double>(S.eta * S.eta) * (1.0 - static_cast<double>(c1 * c1)));
        if (S.tirPossible && cs2 > 1.0f) {
            ++tirCount;  // ld->totalInternalReflection++, zoic.cpp:1135/1142
            return false;
        }
        float k = static_cast<float>(static_cast<double>(S.eta * c1) - sqrt(fabs(1.0 - static_cast<double>(cs2))));
        d = V3{u.x * S.eta + N.x * k, u.y * S.eta + N.y * k, u.z * S.eta + N.z * k};
    }
    return true;
}
ZOIC_HD bool trace_lens_strict(const KolbTable &T, V3 &o, V3 &d, uint32_t &tirCount)
{
    return trace_lens_strict_range(T, o, d, tirCount, 0, T.lensCount - 1);
}

// The interface-0 half of trace_lens_strict, operation for operation: does the ray hit the rear sphere and clear its
// housing (zoic.cpp:1107-1117 for i == 0)?  A `false` here is exactly a `return false` of the first loop iteration.
ZOIC_HD bool interface0_clear_strict(const KolbTable &T, V3 o, V3 d)
{
    const Surface S = T.surf[0];
    V3 u = normalize3(d);
    V3 L{0.0f - o.x, 0.0f - o.y, S.center - o.z};
    float tca = dot3(L, u);
    float d2 = dot3(L, L) - (tca * tca);
    if (d2 > S.radius2) return false;
    float thc = ZOIC_SQRT_RN(fabsf(S.radius2 - d2));
    float t = tca + thc * S.sign;
    V3 hit{o.x + u.x * t, o.y + u.y * t, o.z + u.z * t};
    float h2 = hit.x * hit.x + hit.y * hit.y;
    return !(h2 > S.housing2);
}

// ---- imageData::bokehSample, zoic.cpp:420-485 --------------------------------------------------------
// std::upper_bound: index of the first element > v in a non-decreasing array
ZOIC_HD int upper_bound_idx(const float *a, int n, float v)
{
    int lo = 0, len = n;
    while (len > 0) {
        int half = len >> 1;
        if (!(v < a[lo + half])) {
            lo += half + 1;
            len -= half + 1;
        } else {
            len = half;
        }
    }
    return lo;
}

ZOIC_HD V2 bokeh_sample(const float *cdfRow, const int32_t *rowIndices, const float *cdfColumn, const int32_t *columnIndices,
                        int x, int y, float uRow, float uCol)
{
    int r = upper_bound_idx(cdfRow, y, uRow);
    if (r >= y) r = y - 1;
    int row = rowIndices[r];
    int start = row * x;
    int c = upper_bound_idx(cdfColumn + start, x, uCol);
    if (c >= x) c = x - 1;
    int col = columnIndices[start + c] - start;
    // centring swaps x and y and uses integer division (zoic.cpp:441, 466); *2.0 in f64 is an exact doubling
    float flippedRow = static_cast<float>(col - ((y - 1) / 2));
    float flippedColumn = static_cast<float>(row - ((x - 1) / 2)) * -1.0f;
    return V2{(flippedRow / static_cast<float>(x)) * 2.0f, (flippedColumn / static_cast<float>(y)) * 2.0f};
}

// ---- exit-pupil LUT transform, zoic.cpp:1891-1911 -------------------------------------------------
// keys are 0.125*k (exitPupilLUT: filmWidth 4.0 / 32).  Returns false when the sample is outside the table
// (lower_bound()==end() is dereferenced in the reference: UB, fenced -> maxScale = translation = 0).
ZOIC_HD bool lut_lookup(const KolbTable &T, float dist, float &maxScale, float &translation)
{
    const float samplingErrorCorrection = 1.05f;
    float scaled = dist * 8.0f;                   // exact
    int low = static_cast<int>(ceilf(scaled));    // std::map::lower_bound: first key >= dist
    if (!(scaled <= static_cast<float>(T.lutSize - 1))) {
        maxScale = 0.0f; translation = 0.0f;
        return false;
    }
    if (low <= 0) {
        // dist == 0: `--low` on begin() is UB in the reference (zoic.cpp:1905); use its own d==0 branch of
        // testAperturesLUT (zoic.cpp:1512-1518): entry 0, no interpolation
        maxScale = T.lutMaxScale[0] * samplingErrorCorrection;
        translation = T.lutCentroidX[0];
        return true;
    }
    float lowerBound = static_cast<float>(low) * 0.125f;
    // (dist - lowerBound) / (prev - lowerBound): the divisor is exactly -0.125, so the quotient is the exact product
    float percentage = (dist - lowerBound) * -8.0f;
    float a = T.lutMaxScale[low], b = T.lutMaxScale[low - 1];
    maxScale = (a + percentage * (b - a)) * samplingErrorCorrection;     // linearInterpolate, zoic.cpp:655-657
    float ca = T.lutCentroidX[low], cb = T.lutCentroidX[low - 1];
    translation = ca + percentage * (cb - ca);
    return true;
}

}  // namespace zoic
