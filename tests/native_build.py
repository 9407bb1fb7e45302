"""Sanitizer builds of the product's host C++ and of tests/native/boundary_stress.cpp (no pytest needed: __graft_entry__.build()
prebuilds them, tests/test_sanitizers.py runs them)."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "native", "build")
SAN = {"tsan": ["-fsanitize=thread"], "asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]}


def build(kind):
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
        cxx = _clangxx()   # the compiler hipcc drives: same sanitizer runtime as the library
        subprocess.check_call([cxx, "-std=c++17", "-O1", "-g"] + SAN[kind] + ["-I" + os.path.join(ROOT, "include"), drv_src,
                               "-o", exe, "-L" + OUT, "-l:" + os.path.basename(lib), "-Wl,-rpath," + OUT, "-lpthread"])
    return exe


def _clangxx():
    """clang++ of the ROCm installation hipcc belongs to (hipcc --version prints its InstalledDir)."""
    from zoic_amd import build as zbuild
    hipcc = zbuild._hipcc()
    try:
        for line in subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.splitlines():
            if line.startswith("InstalledDir:"):
                cand = os.path.join(line.split(":", 1)[1].strip(), "clang++")
                if os.path.exists(cand):
                    return cand
    except OSError:
        pass
    for cand in (os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin", "clang++"),
                 "/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("clang++ of the ROCm toolchain not found")
