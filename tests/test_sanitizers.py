"""ThreadSanitizer / AddressSanitizer+UBSan builds of the product's host C++ (capi.cpp, lens_system.cpp and the host side of
the .hip files), driven through the C-ABI by tests/native/boundary_stress.cpp (the reference has no race
detection; its camera_create_ray races on xor128 and the counters).  Without a GPU the driver runs node_update's host
precompute on several tables-only cameras at once; on the GPU box the same binaries also run 16 render threads on ONE camera.
"""
import os
import subprocess

import pytest

from native_build import OUT, ROOT, build as _build


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
    r = _run(exe, [2, 16, 150], kind)
    assert r.returncode == 0, r.stdout[-8000:]
    assert "part 2: 16 render threads x 150 calls on one camera, failures 0" in r.stdout
    assert "part 3: one frame over 3 lanes of device 0, failures 0" in r.stdout
