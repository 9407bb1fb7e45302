"""arnold/zoic_amd_node.cpp: no Arnold SDK exists in this image, so what can be checked here is that the
guarded translation unit compiles and links against libzoic_amd.so, and that every C-ABI call the shim makes is one the
header declares.  The SDK-side macros are unverified until the file meets a real SDK (said so in the file)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_stub_builds_without_the_sdk(tmp_path):
    out = tmp_path / "zoic_amd_stub.so"
    subprocess.check_call(["g++", "-std=c++11", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "arnold", "zoic_amd_node.cpp"), "-o", str(out)])
    syms = subprocess.check_output(["nm", "-D", "--defined-only", str(out)], text=True)
    assert "zoic_amd_arnold_shim_status" in syms


def test_shim_calls_only_declared_entry_points():
    src = open(os.path.join(ROOT, "arnold", "zoic_amd_node.cpp")).read()
    src = re.sub(r"//.*", "", src)
    hdr = open(os.path.join(ROOT, "include", "zoic_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(zoic_[a-z0-9_]+)\s*\(", hdr))
    used = set(re.findall(r"\b(zoic_[a-z0-9_]+)\s*\(", src)) - {"zoic_amd_arnold_shim_status"}
    assert used and used <= declared, used - declared
    # the five node methods + reverse ray + loader of the reference's method table are all mapped
    for method in ("node_parameters", "node_initialize", "node_update", "node_finish", "camera_create_ray", "camera_reverse_ray", "node_loader"):
        assert re.search(r"^%s\b" % method, src, flags=re.M), method
    # the 14 parameter names of zoic.cpp:1547-1562
    for name in ("sensorWidth", "sensorHeight", "focalLength", "fStop", "focalDistance", "useImage", "bokehPath", "lensModel",
                 "lensDataPath", "kolbSamplingLUT", "useDof", "opticalVignettingDistance", "opticalVignettingRadius", "exposureControl"):
        assert '"%s"' % name in src, name


def test_tile_buffer_header_is_plain_cpp11_over_the_c_abi(tmp_path):
    """arnold/zoic_tile_buffer.hpp (accumulate -> flush -> serve over zoic_tile_*) needs no SDK: it compiles on its own as C++11 and
    calls only declared entry points (its GPU test is tests/test_tile_gpu.py::test_the_cpp_tile_buffer_accumulate_flush_serve)."""
    src = tmp_path / "use.cpp"
    src.write_text('#include "zoic_tile_buffer.hpp"\nint use(zoic_camera *c) { ZoicTileBuffer t(c, 64, 0); t.push(0, 0, 0.5f, 0.5f); return (int)t.size(); }\n')
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "arnold"), str(src)])
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "zoic_amd.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(zoic_[a-z0-9_]+)\s*\(", hdr))
    code = re.sub(r"//.*", "", open(os.path.join(ROOT, "arnold", "zoic_tile_buffer.hpp")).read())
    used = set(re.findall(r"\b(zoic_[a-z0-9_]+)\s*\(", code))
    assert used and used <= declared, used - declared


def test_tile_buffer_push_at_capacity_stores_nothing_under_asan(tmp_path):
    """VERDICT r5 #5 / ADVICE r5: round 5's ZoicTileBuffer::push wrote in_[n_] with no capacity check.  tests/native/tile_buffer_overflow.cpp
    drives the header against a MOCK of the zoic_tile_* entry points (exact-size malloc'd arrays, no GPU) under AddressSanitizer: with the
    old header it is a heap-buffer-overflow report, with this one push() returns kFull and stores nothing -- in all four layouts
    (AtCameraInput rows / 16-byte samples in, AtCameraOutput rows / zoic_ray records out)."""
    exe = tmp_path / "tile_buffer_overflow"
    subprocess.check_call(["g++", "-std=c++11", "-g", "-O1", "-Wall", "-fsanitize=address", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "tile_buffer_overflow.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "tile_buffer_overflow OK" in out.stdout, (out.stdout[-300:], out.stderr[-1500:])
