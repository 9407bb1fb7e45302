"""C-ABI argument and state errors on a real device: every misuse returns a status code, never crashes, never falls back."""
import ctypes as C

import numpy as np
import pytest

from zoic_amd import ZoicCamera, ZoicError, _capi
from zoic_amd.workloads import camera_params

pytestmark = pytest.mark.gpu


def test_rays_before_update_is_an_error(gpu):
    cam = ZoicCamera(0)
    with pytest.raises(ZoicError) as e:
        cam.create_rays(np.zeros((4, 4), np.float32))
    assert e.value.status_name == "ZOIC_ERR_NOT_UPDATED"
    # a failed update leaves the camera unusable until a good one
    with pytest.raises(ZoicError):
        cam.update(lensDataPath="/nonexistent.dat")
    with pytest.raises(ZoicError) as e:
        cam.create_rays(np.zeros((4, 4), np.float32))
    assert e.value.status_name == "ZOIC_ERR_NOT_UPDATED"
    cam.update(**camera_params("C2"))
    assert cam.create_rays(np.zeros((4, 4), np.float32))["planes"].shape == (7, 4)


def test_pointer_validation(gpu):
    lib = _capi.load()
    cam = ZoicCamera(0).update(**camera_params("C2"))
    h = cam._h
    import torch
    s = torch.zeros((64, 4), device="cuda")
    r = torch.zeros((64, 8), device="cuda")
    assert lib.zoic_create_rays_device(h, 64, None, None, 0, r.data_ptr(), None) == 1          # NULL samples
    assert lib.zoic_create_rays_device(h, 64, s.data_ptr(), None, 0, None, None) == 1          # NULL rays
    assert lib.zoic_create_rays_device(h, 16, s.data_ptr() + 4, None, 0, r.data_ptr(), None) == 1   # misaligned samples
    assert lib.zoic_create_rays_device(h, 16, s.data_ptr(), None, 0, r.data_ptr() + 8, None) == 1   # misaligned rays
    assert b"aligned" in lib.zoic_last_error_string()
    assert lib.zoic_create_rays_device(h, 0, None, None, 0, None, None) == 0                    # empty batch is fine
    assert lib.zoic_create_rays_device(None, 1, s.data_ptr(), None, 0, r.data_ptr(), None) == 1
    assert lib.zoic_camera_update(h, None) == 1
    assert lib.zoic_camera_set_precision(h, 7) == 1
    hh = C.c_void_p()
    assert lib.zoic_camera_create(99, C.byref(hh)) == 1 and not hh.value                        # no such device
    assert lib.zoic_camera_create(0, None) == 1
    lib.zoic_camera_destroy(None)                                                                # no-op


def test_lens_model_none_produces_no_rays(gpu):
    cam = ZoicCamera(0).update(lensModel=2)   # NONE: camera_create_ray falls through (zoic.cpp:1966-1968)
    with pytest.raises(ZoicError) as e:
        cam.create_rays(np.zeros((4, 4), np.float32))
    assert e.value.status_name == "ZOIC_ERR_INVALID_ARGUMENT"


def test_counters_reset_and_accumulate(gpu):
    cam = ZoicCamera(0).update(**camera_params("C2"))
    tir0 = cam.counters()["totalInternalReflection"]
    s = np.random.RandomState(0).rand(1000, 4).astype(np.float32)
    cam.create_rays(s)
    cam.create_rays(s)
    c = cam.counters()
    assert c["succesRays"] + c["vignettedRays"] == 2000 and c["totalInternalReflection"] >= tir0
    cam.reset_counters()
    assert cam.counters() == dict(succesRays=0, vignettedRays=0, totalInternalReflection=0)


def test_seed_changes_only_retried_rays(gpu):
    cam = ZoicCamera(0).update(**camera_params("C2"))
    rs = np.random.RandomState(1)
    s = np.stack([rs.uniform(-1, 1, 20000), rs.uniform(-0.56, 0.56, 20000), rs.rand(20000), rs.rand(20000)], 1).astype(np.float32)
    a = cam.create_rays(s)
    cam.set_seed(12345)
    b = cam.create_rays(s)
    first = (a["flags"] & 1) == 0
    assert np.array_equal(a["planes"][:, first].view(np.uint32), b["planes"][:, first].view(np.uint32))
    assert (a["planes"][:, ~first] != b["planes"][:, ~first]).any()
    cam.set_seed(1)
    assert np.array_equal(cam.create_rays(s)["planes"].view(np.uint32), a["planes"].view(np.uint32))
