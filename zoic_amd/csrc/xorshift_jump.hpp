// xorshift_jump.hpp -- jump-ahead of the reference's xorshift128 (xor128, zoic.cpp:647-652).
//
// One draw is a linear map of the 128-bit state over GF(2) (shifts and xors only), so n draws are the matrix power M^n and
// "the state after n draws" is one matrix-vector product -- what lets every GPU thread of the exit-pupil LUT build start
// in the middle of the reference's ONE sequential stream (lut_build.hip).  Matrices are stored by columns: M v = the XOR of
// the columns whose bit is set in v (bit k of the state = bit k % 32 of word k / 32, words in the order x, y, z, w).
#pragma once
#include <cstdint>

#include "optics.hpp"

namespace zoic {

struct BitMat128 { uint32_t col[128][4]; };

ZOIC_HD void bitmat_apply(const BitMat128 &m, const uint32_t v[4], uint32_t out[4])
{
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    for (int k = 0; k < 128; ++k) {
        const uint32_t on = 0u - ((v[k >> 5] >> (k & 31)) & 1u);   // all ones when bit k is set
        r0 ^= m.col[k][0] & on; r1 ^= m.col[k][1] & on; r2 ^= m.col[k][2] & on; r3 ^= m.col[k][3] & on;
    }
    out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3;
}

inline Rng bitmat_apply(const BitMat128 &m, const Rng &s)
{
    const uint32_t v[4] = {s.x, s.y, s.z, s.w};
    uint32_t o[4];
    bitmat_apply(m, v, o);
    return Rng{o[0], o[1], o[2], o[3]};
}

inline BitMat128 bitmat_identity()
{
    BitMat128 m{};
    for (int k = 0; k < 128; ++k) m.col[k][k >> 5] = 1u << (k & 31);
    return m;
}

inline BitMat128 bitmat_mul(const BitMat128 &a, const BitMat128 &b)   // a b: apply b first
{
    BitMat128 r;
    for (int k = 0; k < 128; ++k) bitmat_apply(a, b.col[k], r.col[k]);
    return r;
}

inline BitMat128 xor128_step_matrix()   // column k = one draw applied to the basis state e_k
{
    BitMat128 m;
    for (int k = 0; k < 128; ++k) {
        Rng s{0, 0, 0, 0};
        uint32_t *w[4] = {&s.x, &s.y, &s.z, &s.w};
        *w[k >> 5] = 1u << (k & 31);
        (void)xor128(s);
        m.col[k][0] = s.x; m.col[k][1] = s.y; m.col[k][2] = s.z; m.col[k][3] = s.w;
    }
    return m;
}

inline BitMat128 bitmat_pow(BitMat128 base, uint64_t n)
{
    BitMat128 r = bitmat_identity();
    while (n) {
        if (n & 1u) r = bitmat_mul(base, r);
        n >>= 1;
        if (n) base = bitmat_mul(base, base);
    }
    return r;
}

}  // namespace zoic
