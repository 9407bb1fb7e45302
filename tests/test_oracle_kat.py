"""Pins the CPU oracle against the reference's only committed known-answer data: src/draw.zoic.

The fixture tests/golden/draw_zoic_kat.json holds the numbers printed in that dump (header + RAYS) and, per ray, the
three inputs the dump omits (recovered by tests/golden/make_draw_zoic_fixture.py).  Nothing here needs a GPU.
"""
import json
import os

import numpy as np
import pytest

from zoic_amd import lens_path

GOLD = os.path.join(os.path.dirname(__file__), "golden", "draw_zoic_kat.json")


@pytest.fixture(scope="module")
def kat():
    return json.load(open(GOLD))


@pytest.fixture(scope="module")
def cam(oracle_lib, kat):
    c = oracle_lib.OracleCamera()
    cfg = kat["config"]
    c.update(lensDataPath=lens_path(cfg["lens"]), focalLength=cfg["focalLength"], fStop=cfg["fStop"],
             focalDistance=cfg["focalDistance"], sensorWidth=cfg["sensorWidth"], kolbSamplingLUT=False)
    return c


def fmt(v):
    return "%.10f" % float(v)


def test_header_lens_geometry_to_10_decimals(cam, kat):
    """LENSES{} = (-center, -curvature, half-angle) per surface, zoic.cpp:1241-1251: pins parse, mm->cm, the
    focal-length trace, the rescale and computeLensCenters."""
    lt = cam.lens_table()
    el = lt["elements"]
    want = np.array(kat["header"]["LENSES"]).reshape(-1, 3)
    assert lt["lensCount"] == len(want) == 11
    for i in range(len(want)):
        assert fmt(-el[i, 4]) == want[i, 0], "center %d" % i
        assert fmt(-el[i, 0]) == want[i, 1], "curvature %d" % i
        # std::asin((aperture*0.5)/curvature) * (180/AI_PI): f64 asin, f32 factor (zoic.cpp:1248)
        ang = np.arcsin((np.float64(el[i, 3]) * 0.5) / np.float64(el[i, 0])) * np.float64(np.float32(180.0) / np.float32(np.pi))
        assert fmt(ang) == want[i, 2], "half angle %d" % i


def test_header_ior_and_scalars(cam, kat):
    lt = cam.lens_table()
    h = kat["header"]
    assert [fmt(v) for v in lt["elements"][:, 2]] == h["IOR"]
    assert str(lt["apertureElement"]) == h["APERTUREELEMENT"][0]
    assert fmt(-lt["apertureDistance"]) == h["APERTUREDISTANCE"][0]
    assert fmt(lt["userApertureRadius"]) == h["APERTURE"][0]          # kolbFocalLength/(2*fStop), zoic.cpp:1664
    assert fmt(lt["elements"][:, 3].max()) == h["APERTUREMAX"][0]
    assert fmt(-lt["originShift"]) == h["IMAGEDISTANCE"][0]           # calculateImageDistance, zoic.cpp:1054-1095
    assert fmt(-kat["config"]["focalDistance"]) == h["FOCUSDISTANCE"][0]


EXACT_RAYS_OBSERVED = 26      # of the 95 replayable rays (109 complete ones in the dump); the other 69 agree to the last printed place


def test_rays_replay(cam, kat):
    """Every complete ray of RAYS{} whose missing inputs could be recovered is replayed through the oracle's
    traceThroughLensElements: 11 hit points (y,z) + the exit direction (y,z) against the printed values."""
    rays = kat["rays"]
    assert len(rays) >= 90 and kat["n_complete_rays"] == 109
    worst_hit = worst_dir = 0.0
    exact = 0
    for r in rays:
        p = np.array(r["printed"]).reshape(12, 4)
        oz, oy = -p[0, 0], -p[0, 1]
        ox, dx, dy = r["fit_ox_dx_dy"]
        ok, hits, _o, d = cam.trace_record((ox, oy, oz), (dx, dy, r["dir_z"]))
        assert ok and len(hits) == 11
        ref_y, ref_z = -p[:11, 3], -p[:11, 2]
        e = max(np.abs(hits[:, 1] - ref_y).max(), np.abs(hits[:, 2] - ref_z).max())
        # zoic.cpp:1150-1151: hit.z + dir.z * -10000.0 printed in f64
        ref_dz = (float(np.float32(-p[11, 0])) - p[11, 2]) / 10000.0
        ref_dy = (float(np.float32(-p[11, 1])) - p[11, 3]) / 10000.0
        ed = max(abs(float(d[1]) - ref_dy), abs(float(d[2]) - ref_dz))
        worst_hit, worst_dir = max(worst_hit, e), max(worst_dir, ed)
        if all(fmt(-hits[i, 2]) == fmt(p[i, 2]) and fmt(-hits[i, 1]) == fmt(p[i, 3]) for i in range(11)):
            exact += 1
    print("draw.zoic replay: %d rays, %d reproduce all 22 printed hit-point numbers digit for digit; worst hit %.3g, worst direction %.3g"
          % (len(rays), exact, worst_hit, worst_dir))
    assert worst_hit < 1e-6, worst_hit       # one f32 ulp at |z| ~ 8 is 9.5e-7
    assert worst_dir < 1e-6, worst_dir
    # How strong the pin is, stated as a number: the three unprinted inputs of a ray (origin.x, dir.x, dir.y) are FITTED to its 24
    # printed numbers (tests/golden/make_draw_zoic_fixture.py), so a ray that comes back digit for digit pins the 11-interface
    # arithmetic on 21 residual degrees of freedom; the others agree to the last printed place (1e-6: half an f32 ulp at |z| ~ 8),
    # the fit's own residual.  The count is asserted as observed so that an edit of the trace that loses a digit shows up.
    assert exact == EXACT_RAYS_OBSERVED, exact


def test_xor128_known_answers(oracle_lib):
    """Marsaglia xorshift128 with the reference's seed (zoic.cpp:648): first outputs of the published generator."""
    got = oracle_lib.xor128_stream(4)
    assert list(map(int, got)) == [3701687786, 458299110, 2500872618, 3633119408]
