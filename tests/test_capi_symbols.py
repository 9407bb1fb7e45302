"""The C-ABI library loads and exports every symbol include/zoic_amd.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re
import subprocess

from zoic_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "zoic_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zoic_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    names = declared_symbols()
    assert len(names) >= 20
    lib = ctypes.CDLL(_capi.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python binding table covers exactly the header
    assert sorted(_capi.SYMBOLS) == names


def test_dynamic_symbol_table_is_plain_c():
    out = subprocess.check_output(["nm", "-D", "--defined-only", _capi.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for n in declared_symbols():
        assert n in exported, n


def test_library_basics_without_gpu():
    lib = _capi.load()
    assert lib.zoic_abi_version() == _capi.ABI_VERSION == 5
    assert lib.zoic_status_string(0) == b"ZOIC_OK"
    assert lib.zoic_status_string(11) == b"ZOIC_ERR_NO_DEVICE"
    p = _capi.Params()
    lib.zoic_params_default(ctypes.byref(p))
    # node_parameters defaults, zoic.cpp:1547-1562
    assert (round(p.sensorWidth, 4), round(p.sensorHeight, 4), p.focalLength, p.fStop, p.focalDistance) == (3.6, 2.4, 2.0, 4.0, 100.0)
    assert (p.useImage, p.lensModel, p.kolbSamplingLUT, p.useDof) == (0, 1, 1, 1)
    assert (p.opticalVignettingDistance, p.opticalVignettingRadius, p.exposureControl) == (0.0, 1.0, 0.0)
    assert ctypes.sizeof(_capi.CameraInput) == 28 and ctypes.sizeof(_capi.CameraOutput) == 84


def test_no_device_is_loud():
    """Without a GPU the product refuses to create a ray-producing camera: there is no CPU fallback."""
    lib = _capi.load()
    if lib.zoic_device_count() > 0:
        return
    h = ctypes.c_void_p()
    assert lib.zoic_camera_create(0, ctypes.byref(h)) == 11
    assert b"no CPU path" in lib.zoic_last_error_string()


def test_product_never_touches_the_oracle():
    """The product path must not import, link or dlopen anything under oracle/."""
    for dirpath, _dirs, files in os.walk(os.path.join(ROOT, "zoic_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src and "zoic_oracle" not in src, f
    deps = subprocess.check_output(["ldd", _capi.LIB_PATH], text=True)
    assert "oracle" not in deps
