"""tools/region_times.py [C3] [fast|strict] -- where do the waves of the refill kernel spend their time?
Builds a DEBUG copy of the library with -DZOIC_REGION_TIMERS (s_memtime at the region boundaries of the pass loop,
summed per wave), runs one frame and prints the share of wave-cycles per region.  Debug tool, not the product path."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zoic_amd import build as B
dbg = os.path.join(ROOT, "tools", "ubench", "libzoic_amd_rt.so")
if "--build" in sys.argv or not os.path.exists(dbg):
    cmd = [B._hipcc()] + B.FLAGS + ["-DZOIC_REGION_TIMERS"] + os.environ.get("ZOIC_EXTRA_HIPCC_FLAGS", "").split() + [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-o", dbg]
    subprocess.check_call(cmd)
    if "--build" in sys.argv:
        sys.exit(0)
import zoic_amd._capi as capi
capi.LIB_PATH = dbg
import torch
from zoic_amd import ZoicCamera, workloads
args = [a for a in sys.argv[1:] if not a.startswith("--")]
cfg = args[0] if args else "C3"
prec = args[1] if len(args) > 1 else "fast"
from zoic_amd import PRECISION_FAST, PRECISION_STRICT
C_ = workloads.CONFIGS[cfg]
cam = ZoicCamera(device=0)
if C_["bokeh"]:
    cam.set_bokeh_image(workloads.hexagon_bokeh())
cam.update(**workloads.camera_params(cfg))
cam.set_precision(PRECISION_FAST if prec == "fast" else PRECISION_STRICT)
n = workloads.ray_count(cfg)
s = cam.generate_samples(n, C_["width"], C_["height"], C_["spp"], seed=1, ray_index_base=0)
lib = capi.load()
lib.zoic_debug_region_cycles.restype = C.c_int
lib.zoic_debug_region_cycles.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 8)()
cam.create_rays(s); torch.cuda.synchronize()
lib.zoic_debug_region_cycles(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); cam.create_rays(s); e1.record(); torch.cuda.synchronize()
lib.zoic_debug_region_cycles(buf, 1)
names = ["refill", "search(sample+pretest)", "trace", "(unused)", "finish/store"]
tot = sum(buf[i] for i in range(5))
print("%s %s: %d rays, %.3f ms with timers, %d waves, %.0f s_memtime ticks/wave" % (cfg, prec, n, e0.elapsed_time(e1), buf[7], tot / max(buf[7], 1)))
for i, nm in enumerate(names):
    print("  %-24s %5.1f %%" % (nm, 100.0 * buf[i] / tot))
