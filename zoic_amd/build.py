"""Build libzoic_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m zoic_amd.build [--force]

The shared library is the product: HIP kernels + the C-ABI of include/zoic_amd.h.  It depends on
libamdhip64 only (no torch, no Python).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzoic_amd.so")
SOURCES = ["capi.cpp", "frame.cpp", "lens_system.cpp", "kernels.hip", "kolb_pool.hip", "kolb_pool_dead.hip", "kolb_pool_two.hip", "kolb_listed.hip", "thin_refill.hip", "bokeh_cdf.hip", "mailbox.hip", "lut_build.hip"]
# every header of csrc/ is a dependency of every object (a hand-kept list went stale twice: exact_math.hpp, kolb_refill_body.hpp)
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join(ROOT, "include", "zoic_amd.h")]

# -ffp-contract=off: strict kernels and the host precompute must round exactly like the CPU oracle;
# the FAST arithmetic is written with explicit FMAs (csrc/fast_optics.hpp: contraction stays off there too).  -fno-slp-vectorize: packing scalar f32 math into
# v_pk_* costs more in register shuffles than it saves on gfx950 (measured +8..20 % Mrays/s without it).
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libzoic_amd.so cannot be built (no CPU fallback exists)")


OBJDIR = os.path.join(HERE, "build")


def _deps():
    return [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in [os.path.join(CSRC, s) for s in SOURCES] + _deps())


def build(force=False, verbose=False, extra_flags=(), out=None, objdir=None):
    """Compile every source to an object (in parallel, recompiling only what changed) and link the shared library."""
    out = out or LIB
    extra_flags = list(extra_flags) + os.environ.get("ZOIC_EXTRA_HIPCC_FLAGS", "").split()
    if out == LIB and not extra_flags and not force and not needs_build():
        return LIB
    import zlib
    objdir = objdir or (OBJDIR if not extra_flags else OBJDIR + "_" + "%08x" % zlib.crc32(" ".join(extra_flags).encode()))
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    cflags = [f for f in FLAGS if f != "-shared"] + extra_flags
    # objects are only reused under the flags they were compiled with (the stamp also catches an edited FLAGS list)
    stamp = os.path.join(objdir, "flags.stamp")
    flagline = " ".join(cflags)
    if not os.path.exists(stamp) or open(stamp).read() != flagline:
        force = True
        with open(stamp, "w") as f:
            f.write(flagline)
    newest_header = max(os.path.getmtime(d) for d in _deps())
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        srcp = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(srcp), newest_header):
            continue
        cmd = [hipcc] + cflags + ["-c", srcp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed on: %s" % ", ".join(failed))
    link_extra = [f for f in extra_flags if f.startswith("-fsanitize") or f == "-g"]
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950"] + link_extra + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
