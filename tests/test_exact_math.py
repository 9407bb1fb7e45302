"""The integer forms the STRICT kernels use for 1 / sqrt(s) of an already-unit vector (csrc/exact_math.hpp
rcp_sqrt_rn_near_one): both roundings -- sqrtss, then divss, what the reference's AiV3Normalize does -- are exact integer
functions of the bits of s while s is 1 to within 2048 ulps.  Checked here against IEEE sqrt and divide (numpy on x86: correctly
rounded) on EVERY value of the range; the formula is pure integer arithmetic, so the GPU computes the same bits."""
import numpy as np

ONE = 0x3F800000
RANGE = 2048      # kUnitRange


def test_sqrt_and_reciprocal_of_a_nearly_unit_length_are_integer_functions_of_its_bits():
    i = np.arange(-RANGE, RANGE + 1, dtype=np.int64)
    s = (ONE + i).astype(np.uint32).view(np.float32)
    y = np.sqrt(s)                                      # sqrtss
    t = (np.float32(1.0) / y).astype(np.float32)        # divss
    assert np.array_equal(y.view(np.uint32).astype(np.int64), ((ONE + i) + ONE) >> 1)
    assert np.array_equal(t.view(np.uint32).astype(np.int64), ONE - np.maximum(i & ~1, i >> 2))


def test_the_range_is_not_generous():
    """The reciprocal's formula really ends: first failure at i = 2898 (the second-order term crosses a rounding boundary)."""
    i = np.arange(2049, 4097, dtype=np.int64)
    s = (ONE + i).astype(np.uint32).view(np.float32)
    t = (np.float32(1.0) / np.sqrt(s)).astype(np.float32)
    bad = i[t.view(np.uint32).astype(np.int64) != ONE - np.maximum(i & ~1, i >> 2)]
    assert bad.size and bad.min() == 2898


def test_cs2_is_one_fma_away_from_grazing_incidence():
    """calculateTransmissionVector's cs2 = (float)((double)(eta*eta) * (1.0 - (double)(c1*c1))) (zoic.cpp:1016) equals
    fmaf(-eta2, c1sq, eta2) whenever c1sq >= 1/32: 1 - c1sq then has at most 29 significant bits, its f64 product with the 24-bit
    eta2 is exact, and the conversion to float is the single rounding of eta2 - eta2 c1sq.  (csrc/kolb_device.hpp uses the fma
    there and the reference's expression below 1/32.)  Exact rational arithmetic stands in for the fused multiply-add."""
    from fractions import Fraction
    import math

    def rn24(fr):
        if fr == 0:
            return np.float32(0.0)
        sign, a = (-1 if fr < 0 else 1), abs(fr)
        e = math.floor(math.log2(a))
        while Fraction(2) ** (e + 1) <= a:
            e += 1
        while Fraction(2) ** e > a:
            e -= 1
        ulp = Fraction(2) ** (e - 23)
        q = a / ulp
        n = q.numerator // q.denominator
        r = q - n
        if r > Fraction(1, 2) or (r == Fraction(1, 2) and n % 2 == 1):
            n += 1
        return np.float32(sign * float(n * ulp))
    rng = np.random.default_rng(7)
    n = 6000
    eta2 = rng.uniform(0.3, 3.0, n).astype(np.float32)
    c1sq = np.concatenate([rng.uniform(1 / 32, 1.0, n - 300), 1 + rng.integers(-6, 7, 200) * 2.0 ** -23,
                           np.float32(1 / 32) + np.arange(100) * 2.0 ** -28]).astype(np.float32)
    assert c1sq.min() >= np.float32(0.03125)
    for a, p in zip(eta2, c1sq):
        ref = np.float32(np.float64(a) * (1.0 - np.float64(p)))
        assert ref == rn24(Fraction(float(a)) * (1 - Fraction(float(p)))), (a, p)
    # where exactness ends: floats in [1/64, 1/32) are multiples of 2^-29 (29 bits of 1 - p: still exact, the kernel's 1/32 is on the
    # safe side), the first float below 1/64 is a multiple of 2^-30: 30 bits, a 54-bit product
    p = np.float32(1 / 64) - np.float32(2.0 ** -30)
    assert (1 - Fraction(float(p))).numerator.bit_length() == 30
