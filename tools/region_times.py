"""tools/region_times.py [C3] [fast|strict] -- where do the waves of the refill kernel spend their time?
Builds a DEBUG copy of the library with -DZOIC_REGION_TIMERS (s_memtime at the region boundaries of the pass loop,
summed per wave), runs one frame and prints the share of wave-cycles per region.  Debug tool, not the product path."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zoic_amd import build as B
dbg = os.path.join(ROOT, "tools", "ubench", "libzoic_amd_rt.so")
if "--build" in sys.argv or not os.path.exists(dbg):
    B.build(force=True, extra_flags=["-DZOIC_REGION_TIMERS"], out=dbg, objdir=os.path.join(ROOT, "tools", "ubench", "obj_rt"))
    if "--build" in sys.argv:
        sys.exit(0)
import zoic_amd._capi as capi
capi.LIB_PATH = dbg
import torch
from zoic_amd import ZoicCamera, workloads
args = [a for a in sys.argv[1:] if not a.startswith("--")]
cfg = args[0] if args else "C3"
prec = args[1] if len(args) > 1 else "fast"
from zoic_amd import PRECISION_FAST, PRECISION_FAST_UNCHECKED, PRECISION_STRICT
C_ = workloads.CONFIGS[cfg]
cam = ZoicCamera(device=0)
if C_["bokeh"]:
    cam.set_bokeh_image(workloads.hexagon_bokeh())
cam.update(**workloads.camera_params(cfg))
cam.set_precision({"fast": PRECISION_FAST, "unchecked": PRECISION_FAST_UNCHECKED, "strict": PRECISION_STRICT}[prec])
n = workloads.ray_count(cfg)
s = cam.generate_samples(n, C_["width"], C_["height"], C_["spp"], seed=1, ray_index_base=0)
lib = capi.load()
lib.zoic_debug_region_cycles.restype = C.c_int
lib.zoic_debug_region_cycles.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 8)()
ps = (C.c_ulonglong * 8)()
lib.zoic_debug_pass_stats.restype = C.c_int
lib.zoic_debug_pass_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
cam.create_rays(s); torch.cuda.synchronize()
lib.zoic_debug_region_cycles(buf, 1)
lib.zoic_debug_pass_stats(ps, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); cam.create_rays(s); e1.record(); torch.cuda.synchronize()
lib.zoic_debug_region_cycles(buf, 1)
names = ["refill", "search(sample+pretest)", "trace", "(unused)", "finish/store"]
tot = sum(buf[i] for i in range(5))
print("%s %s: %d rays, %.3f ms with timers, %d waves, %.0f s_memtime ticks/wave" % (cfg, prec, n, e0.elapsed_time(e1), buf[7], tot / max(buf[7], 1)))
for i, nm in enumerate(names):
    print("  %-24s %5.1f %%" % (nm, 100.0 * buf[i] / tot))
lib.zoic_debug_pass_stats(ps, 1)
passes, act, it, look, tp, cand = [ps[i] for i in range(6)]
print("  wave-passes %d (%.2f per 64 rays); active lanes/pass %.1f; search iterations/pass %.2f, looking lanes/iteration %.1f; "
      "trace in %.0f %% of passes with %.1f candidate lanes" % (passes, passes * 64.0 / n, act / max(passes, 1), it / max(passes, 1), look / max(it, 1),
                                                                   100.0 * tp / max(passes, 1), cand / max(tp, 1)))
