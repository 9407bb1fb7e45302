"""tools/wave_timeline.py [C2] [mode] [spp]: per-wave start / exhaustion / end times of one launch (debug build, -DZOIC_REGION_TIMERS)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zoic_amd._capi as capi
capi.LIB_PATH = os.path.join(ROOT, "tools", "ubench", "libzoic_timers.so")
import torch
from zoic_amd import ZoicCamera, workloads, PRECISION_FAST, PRECISION_FAST_UNCHECKED, PRECISION_STRICT
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
prec = sys.argv[2] if len(sys.argv) > 2 else "unchecked"
c = workloads.CONFIGS[cfg]
spp = int(sys.argv[3]) if len(sys.argv) > 3 else c["spp"]
cam = ZoicCamera(device=0)
if c["bokeh"]:
    cam.set_bokeh_image(workloads.hexagon_bokeh())
cam.update(**workloads.camera_params(cfg))
cam.set_precision({"fast": PRECISION_FAST, "unchecked": PRECISION_FAST_UNCHECKED, "strict": PRECISION_STRICT}[prec])
n = c["width"] * c["height"] * spp
s = cam.generate_samples(n, c["width"], c["height"], spp, seed=1)
lib = capi.load()
lib.zoic_debug_wave_log.restype = C.c_int
lib.zoic_debug_wave_log.argtypes = [C.c_void_p, C.c_int]
for _ in range(3):
    cam.create_rays(s)
torch.cuda.synchronize()
log = np.zeros((8192, 4), dtype=np.uint64)
for dead in (1, 0):
    assert lib.zoic_debug_wave_log(log.ctypes.data, dead) == 0
    if log[:, 2].max() > 0:
        break
L = log[log[:, 2] > 0].astype(np.int64)
t0 = L[:, 0].min()
st, ex, en, ps = (L[:, 0] - t0) / 100.0, (np.where(L[:, 1] > 0, L[:, 1], L[:, 2]) - t0) / 100.0, (L[:, 2] - t0) / 100.0, L[:, 3]
q = lambda v: " ".join("%7.1f" % x for x in np.percentile(v, [0, 1, 10, 50, 90, 99, 100]))
print("%s %s n=%d: %d waves logged; kernel span %.1f us" % (cfg, prec, n, len(L), en.max()))
print("percentiles          min      1%     10%     50%     90%     99%     max")
print("start      [us]  " + q(st))
print("exhausted  [us]  " + q(ex))
print("end        [us]  " + q(en))
print("tail (end-exh)   " + q(en - ex))
print("passes           " + q(ps))
print("us per pass      " + q((en - st) / np.maximum(ps, 1)))
late = st > 5.0
print("waves that started > 5 us after the first: %d (their passes: %s)" % (late.sum(), q(ps[late]) if late.any() else "-"))
