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
