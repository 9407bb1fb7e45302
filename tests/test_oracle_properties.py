"""Property pins for the oracle functions the reference holds no vectors for (concentricDiskSample, fastSin/fastCos,
bokehSample, the LUT lookup): exact identities that follow from the reference's formulas (zoic.cpp line cited per test).
They do not prove the restatement right -- the line-by-line reading does -- they keep a future edit from drifting silently.
"""
import math

import numpy as np

from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh


def f32(x):
    return np.float32(x)


def test_fast_sin_cos_exact_points(oracle_lib):
    """fastSin (zoic.cpp:661-668): x = fmod(x + pi, 2pi) - pi; y = (4/pi) x - (4/pi^2) x|x|; 0.225 (y|y| - y) + y.
    At 0 the parabola is exactly 0; at +-pi/2 the first parabola gives y = 1 to float rounding and the correction term
    0.225 (y|y| - y) then vanishes to the same rounding."""
    assert oracle_lib.fast_sin(0.0) == 0.0
    assert abs(float(oracle_lib.fast_sin(math.pi / 2)) - 1.0) < 5e-7
    assert abs(float(oracle_lib.fast_sin(-math.pi / 2)) + 1.0) < 5e-7
    # odd symmetry: the formula only has odd terms; x + pi rounds differently for +x and -x, hence a few ulps
    for x in (0.1, 0.7, 1.3, 2.9):
        assert abs(float(oracle_lib.fast_sin(x)) + float(oracle_lib.fast_sin(-x))) < 1e-6
    # fastCos(x) = fastSin-parabola of x + pi/2 (zoic.cpp:671-681)
    assert abs(float(oracle_lib.fast_cos(0.0)) - 1.0) < 5e-7
    assert abs(float(oracle_lib.fast_cos(math.pi / 2))) < 5e-7
    # the approximation error of the corrected parabola is < 1.1e-3 everywhere
    xs = np.linspace(-3.1, 3.1, 4001)
    err = max(abs(float(oracle_lib.fast_sin(float(x))) - math.sin(x)) for x in xs)
    assert err < 1.1e-3


def test_concentric_disk_sample_symmetries(oracle_lib):
    """concentricDiskSample (zoic.cpp:686-704): a = 2u-1, b = 2v-1 (f64, narrowed); |a|>|b|: r=a, phi=(pi/4)(b/a), else r=b,
    phi = pi/2 - (pi/4)(a/b); lens = (r fastCos phi, r fastSin phi)."""
    d = oracle_lib.concentric_disk_sample
    # on the u axis (v = 0.5 -> b = 0): phi = 0, lens = (a * fastCos(0), a * fastSin(0)) = (a * ~1, 0)
    x, y = d(0.75, 0.5)
    assert y == 0.0 and abs(float(x) - 0.5) < 1e-6
    x2, y2 = d(0.25, 0.5)
    assert y2 == 0.0 and x2 == -x              # mirror sample: r flips sign, phi stays 0
    # the centre is 0/0 -> NaN (zoic.cpp:697-699), not a silent zero
    xc, yc = d(0.5, 0.5)
    assert math.isnan(float(xc)) and math.isnan(float(yc))
    # samples stay inside the unit disk (the parabola sin/cos overshoot by < 1.1e-3)
    rng = np.random.default_rng(3)
    for u, v in rng.random((2000, 2)):
        px, py = d(float(u), float(v))
        assert float(px) ** 2 + float(py) ** 2 < 1.0 + 3e-3
    # the corners map to radius 1 at 45 degrees: a = b = +-1 -> r = b, phi = pi/2 - pi/4
    cx, cy = d(1.0, 1.0)
    assert abs(math.hypot(float(cx), float(cy)) - 1.0) < 2e-3 and abs(float(cx) - float(cy)) < 2e-3


def test_bokeh_sample_of_a_one_hot_image_is_that_pixel(oracle_lib):
    """bokehSample (zoic.cpp:420-485) on an image with ONE bright pixel must return that pixel's centre for every (u1, u2),
    with the reference's quirks: x and y swapped in the centring (441, 466) and integer halving of (dim - 1)."""
    w, h = 24, 16
    for (r, c) in ((0, 0), (5, 17), (15, 23), (8, 12)):
        img = np.zeros((h, w, 3), np.float32)
        img[r, c] = 1.0
        oc = oracle_lib.OracleCamera()
        oc.set_bokeh_image(img)
        oc.update(**dict(camera_params("C1"), useImage=True, bokehPath="mem:onehot"))
        expect_dx = f32(f32(c - (h - 1) // 2) / f32(w)) * f32(2.0)        # zoic.cpp:466, 479, 483: column centred with y
        expect_dy = f32(f32(f32(r - (w - 1) // 2) * f32(-1.0)) / f32(h)) * f32(2.0)   # zoic.cpp:441, 480, 484: row centred with x
        for u1, u2 in ((0.0, 0.0), (0.3, 0.9), (0.999, 0.5), (0.99999994, 0.99999994)):
            dx, dy = oc.bokeh_sample(u1, u2)
            assert dx == expect_dx and dy == expect_dy, ((r, c), (u1, u2), dx, dy)
        # u == 1.0 (xor128()/2^32 can round to it, zoic.cpp:1806): std::upper_bound finds no element > 1.0, the index is
        # clamped to the LAST entry of the descending sort -- a zero-mass row/column, not the bright pixel
        t = oc.bokeh_tables()
        last_row = int(t["rowIndices"][-1])
        dx1, dy1 = oc.bokeh_sample(1.0, 1.0)
        assert dy1 == f32(f32(f32(last_row - (w - 1) // 2) * f32(-1.0)) / f32(h)) * f32(2.0)


def test_bokeh_cdf_tables_are_distributions(oracle_lib):
    """bokehProbability (zoic.cpp:222-417): cdfRow ends at ~1, every row of cdfColumn restarts at its first weight and ends at
    ~1 (rows with mass) or stays 0 (empty rows); the index arrays are permutations."""
    oc = oracle_lib.OracleCamera()
    oc.set_bokeh_image(hexagon_bokeh(64))
    oc.update(**dict(camera_params("C1"), useImage=True, bokehPath="mem:hex"))
    t = oc.bokeh_tables()
    x, y = t["x"], t["y"]
    assert abs(float(t["cdfRow"][-1]) - 1.0) < 1e-4 and (np.diff(t["cdfRow"]) >= 0).all()
    assert sorted(t["rowIndices"]) == list(range(y))
    cc = t["cdfColumn"].reshape(y, x)
    ends = cc[:, -1]
    assert ((np.abs(ends - 1.0) < 1e-3) | (ends == 0.0)).all()
    ci = t["columnIndices"].reshape(y, x)
    for r in range(y):
        assert sorted(ci[r] - r * x) == list(range(x))


def test_lut_lookup_is_continuous_across_bins_and_fenced_at_the_ends(oracle_lib):
    """Exit-pupil LUT lookup (zoic.cpp:1891-1911): linear interpolation between the two keys around d -- continuous at the
    keys -- and the two UB cases fenced (d == 0: entry 0; d beyond the last key: weight 0 / flag bit 6)."""
    oc = oracle_lib.OracleCamera()
    oc.update(**camera_params("C2"))
    keys, boxes = oc.lut()
    assert len(keys) == 32 and np.allclose(keys, 0.125 * np.arange(32))
    half = 3.6 * 0.5
    lens = (0.3, 0.6)
    for k in (3, 7, 10):                      # straddle key k on the +x axis: origin.x = sx * sensorWidth/2
        d = float(keys[k])
        eps = 1e-5
        lo = oc.create_rays(np.array([[(d - eps) / half, 0.0, *lens]], np.float32), rng_states=np.ones((1, 4), np.uint32))
        hi = oc.create_rays(np.array([[(d + eps) / half, 0.0, *lens]], np.float32), rng_states=np.ones((1, 4), np.uint32))
        if lo["tries"][0] == 0 and hi["tries"][0] == 0:
            assert np.abs(lo["dir"][:, 0] - hi["dir"][:, 0]).max() < 1e-3
    far = oc.create_rays(np.array([[3.95 / half, 0.0, *lens]], np.float32), rng_states=np.ones((1, 4), np.uint32))
    assert far["weight"][0] == 0.0            # beyond key 31 (3.875 cm): fenced, never dereferences end()
    centre = oc.create_rays(np.array([[0.0, 0.0, *lens]], np.float32), rng_states=np.ones((1, 4), np.uint32))
    assert np.isfinite(centre["dir"]).all()   # d == 0: the reference's own d==0 branch (zoic.cpp:1512-1518)
