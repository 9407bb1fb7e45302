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
