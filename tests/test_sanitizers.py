"""ThreadSanitizer / AddressSanitizer+UBSan builds of the product's host C++ (capi.cpp, lens_system.cpp and the host side of
the .hip files), driven through the C-ABI by tests/native/boundary_stress.cpp (the reference has no race
detection; its camera_create_ray races on xor128 and the counters).  Without a GPU the driver runs node_update's host
precompute on several tables-only cameras at once; on the GPU box the same binaries also run 16 render threads on ONE camera.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "native", "build")
SAN = {"tsan": ["-fsanitize=thread"], "asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]}


def _build(kind):
    from zoic_amd import build as zbuild
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, "libzoic_amd_%s.so" % kind)
    flags = SAN[kind] + ["-g", "-O1"]
    srcs = [os.path.join(zbuild.CSRC, s) for s in zbuild.SOURCES] + [os.path.join(zbuild.CSRC, h) for h in zbuild.HEADERS if not os.path.isabs(h)]
    drv_src = os.path.join(ROOT, "tests", "native", "boundary_stress.cpp")
    exe = os.path.join(OUT, "boundary_stress_%s" % kind)
    newest = max(os.path.getmtime(p) for p in srcs + [drv_src, os.path.join(ROOT, "include", "zoic_amd.h")])
    if not (os.path.exists(lib) and os.path.exists(exe) and min(os.path.getmtime(lib), os.path.getmtime(exe)) > newest):
        zbuild.build(force=False, extra_flags=flags, out=lib, objdir=os.path.join(OUT, "obj_" + kind))
        cxx = "/opt/rocm/lib/llvm/bin/clang++"   # the compiler hipcc drives: same sanitizer runtime as the library
        subprocess.check_call([cxx, "-std=c++17", "-O1", "-g"] + SAN[kind] + ["-I" + os.path.join(ROOT, "include"), drv_src,
                               "-o", exe, "-L" + OUT, "-l:" + os.path.basename(lib), "-Wl,-rpath," + OUT, "-lpthread"])
    return exe


def _run(exe, args, kind):
    env = dict(os.environ)
    env["TSAN_OPTIONS"] = "halt_on_error=0 exitcode=66 suppressions=%s" % os.path.join(ROOT, "tests", "native", "tsan.supp")
    env["ASAN_OPTIONS"] = "detect_leaks=0 exitcode=67"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1"
    return subprocess.run([exe] + [str(a) for a in args], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)


@pytest.mark.parametrize("kind", ["tsan", "asan"])
def test_host_precompute_is_clean_under_sanitizers(kind):
    exe = _build(kind)
    r = _run(exe, [3, 0, 0], kind)
    assert r.returncode == 0, r.stdout[-6000:]
    assert "part 1: 3 tables-only cameras updated concurrently, failures 0" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["tsan", "asan"])
def test_sixteen_render_threads_are_clean_under_sanitizers(gpu, kind):
    exe = os.path.join(OUT, "boundary_stress_%s" % kind)
    if not os.path.exists(exe):     # normally prebuilt by the CPU test / __graft_entry__.build() and shipped with the snapshot
        exe = _build(kind)
    r = _run(exe, [2, 16, 250], kind)
    assert r.returncode == 0, r.stdout[-8000:]
    assert "part 2: 16 render threads x 250 calls on one camera, failures 0" in r.stdout
