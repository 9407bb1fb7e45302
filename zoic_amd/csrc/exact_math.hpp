// exact_math.hpp -- correctly rounded f32 square root and reciprocal for the STRICT kernels, leaner than the sequences
// hipcc emits for sqrtf() and 1.0f/x (18 and 12 instructions: denormal scaling, +-1 ulp probing, div_scale/div_fixup).
//
// STRICT mode must round exactly like the reference's SSE sqrtss / divss.  On the operands this path meets (squared vector
// lengths and lengths of order 1e-6 ... 1e9) one Newton step with FMA on the hardware's 1-ulp estimate is already the
// correctly rounded result; tools/ubench/exact_math_check.hip verifies that EXHAUSTIVELY (all 2^32 bit patterns) against the
// f64 route, for the lean sequence inside its guard range and for the guarded function everywhere.  Outside the range
// (zero, denormals, huge, inf, NaN) the guarded functions take the compiler's standard sequence under a rare branch.
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace zoic {

// valid for kExactLo <= x <= kExactHi
constexpr float kExactLo = 1.0e-30f, kExactHi = 1.0e30f;

__device__ __forceinline__ float sqrt_rn_lean(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);          // 1/sqrt(x), 1 ulp
    const float g = x * y;                             // ~sqrt(x)
    const float h = 0.5f * y;                          // ~1/(2 sqrt(x))
    const float e = __builtin_fmaf(-g, g, x);          // x - g^2 (exact to rounding)
    return __builtin_fmaf(e, h, g);
}

__device__ __forceinline__ float rcp_rn_lean(float x)
{
    const float y = __builtin_amdgcn_rcpf(x);          // 1 ulp
    const float e = __builtin_fmaf(-x, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}

__device__ __forceinline__ float sqrt_rn(float x)
{
    if (__builtin_expect(x >= kExactLo && x <= kExactHi, 1)) return sqrt_rn_lean(x);
    return sqrtf(x);
}

__device__ __forceinline__ float rcp_rn(float x)
{
    const float ax = fabsf(x);
    if (__builtin_expect(ax >= kExactLo && ax <= kExactHi, 1)) return rcp_rn_lean(x);
    return 1.0f / x;
}

// The reference re-normalises vectors that are unit already (the surface normal a second time inside Snell, zoic.cpp:1010; the
// refracted direction at the next interface, zoic.cpp:974): s = |a|^2 is then 1 to a few ulps, and for such s BOTH roundings of
// t = 1 / sqrt(s) -- sqrtss, then divss -- are exact integer functions of the bits of s.  With i = bits(s) - bits(1.0f):
//     bits(sqrt_rn(s))           = (bits(s) + 0x3f800000) >> 1                 (= 1.0f's bits + floor(i / 2): a tie rounds DOWN,
//                                                                                the true root lies below 1 + e/2 by e^2/8)
//     bits(rcp_rn(sqrt_rn(s)))   = 0x3f800000 - max(i & ~1, i >> 2)            (arithmetic shift)
// exact for |i| <= 2048 (|s - 1| < 1.2e-4); tests/test_exact_math.py checks both against IEEE sqrt and divide on every one of
// those 4097 values (and shows the second formula failing from i = 2898 on).  Seven integer instructions instead of
// v_rsq + v_rcp + six FMA-class ones; outside the range the caller takes the lean sequences.
constexpr int kUnitRange = 2048;
__device__ __forceinline__ float rcp_sqrt_rn_near_one(float s, bool &inRange)
{
    const int i = __builtin_bit_cast(int, s) - 0x3f800000;
    inRange = static_cast<uint32_t>(i + kUnitRange) <= static_cast<uint32_t>(2 * kUnitRange);
    const int a = i & ~1, b = i >> 2;
    return __builtin_bit_cast(float, 0x3f800000 - (a > b ? a : b));
}

// sqrt(|1 - cs2|) in f64 for a FLOAT cs2 (calculateTransmissionVector, zoic.cpp:1023: `std::sqrt(std::abs(1.0 - cs2))`):
// the operand is a function of one float, so the whole domain is 2^32 values and tools/ubench/exact_math_check.hip compares
// this sequence with the correctly rounded f64 square root on every one of them.  It is hipcc's own Newton scheme on
// v_rsq_f64 without the denormal scaling (v_ldexp_f64 x2, class test, selects): |1 - cs2| is 0 or >= 2^-24, never denormal.
// valid for s == 0 or kExactLo64 <= s <= kExactHi64 (s = |1 - cs2|)
constexpr double kExactLo64 = 1.0e-30, kExactHi64 = 1.0e30;
__device__ __forceinline__ double sqrt64_rn_lean(double s)
{
    const double y = __builtin_amdgcn_rsq(s);
    double g = s * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, s);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, s);
    g = __builtin_fma(d, h, g);
    return s == 0.0 ? 0.0 : g;
}

}  // namespace zoic

#pragma clang fp contract(off)
