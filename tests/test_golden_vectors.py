"""Committed golden vectors (tests/golden/oracle_vectors.npz, made by make_oracle_vectors.py): the oracle must keep
reproducing them bit for bit (CPU), and the HIP path must match them through the C-ABI (GPU, strict mode)."""
import os

import numpy as np
import pytest

from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.npz"))
CASES = ["C1", "C1ov", "C2", "C3", "C4", "C5"]


def params_for(case):
    p = camera_params(case[:2])
    if case == "C1ov":
        p["opticalVignettingDistance"] = 5.0
    return p


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_golden_vectors(oracle_lib, case):
    oc = oracle_lib.OracleCamera()
    if CONFIGS[case[:2]]["bokeh"]:
        oc.set_bokeh_image(hexagon_bokeh())
    oc.update(**params_for(case))
    r = oc.create_rays(GOLD[case + "_samples"], rng_states=GOLD[case + "_states"])
    assert np.array_equal(r["flags"], GOLD[case + "_flags"])
    assert np.array_equal(r["planes"].view(np.uint32), GOLD[case + "_planes"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_strict_matches_golden_vectors(gpu, case):
    from zoic_amd import ZoicCamera
    cam = ZoicCamera(0)
    if CONFIGS[case[:2]]["bokeh"]:
        cam.set_bokeh_image(hexagon_bokeh())
    cam.update(**params_for(case))
    r = cam.create_rays(GOLD[case + "_samples"], rng_states=GOLD[case + "_states"])
    assert np.array_equal(r["flags"], GOLD[case + "_flags"])
    assert np.array_equal(r["planes"].view(np.uint32), GOLD[case + "_planes"].view(np.uint32))
