"""Which section of a pass pays for a slow pair of buffers?  Debug build with per-section cycle counters
(tools/build_variant.sh stats -DZOIC_PASS_STATS), C3 decision-safe, every pair of 3 sample x 3 ray buffers.
    ZOIC_AMD_LIB=$PWD/tools/ubench/libzoic_stats.so python tools/r6_pair_sections.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from zoic_amd import _capi
from zoic_amd.workloads import CONFIGS, ray_count

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
prec = sys.argv[2] if len(sys.argv) > 2 else "fast"
cfg = CONFIGS[name]
n = ray_count(name)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = _capi.load()
cam = bench.make_camera(name, prec, 0)
K = 3
pads, samples, outs = [], [], []
s0 = cam.generate_samples(n, cfg["width"], cfg["height"], cfg["spp"], seed=1, ray_index_base=0)
for i in range(K):
    pads.append(torch.empty((7 + 11 * i) * 1024 * 1024 + 8192, dtype=torch.uint8, device=dev))
    samples.append(s0 if i == 0 else s0.clone())
    pads.append(torch.empty((5 + 3 * i) * 1024 * 1024 + 4096, dtype=torch.uint8, device=dev))
    outs.append(dict(rays=torch.empty((n, 8), dtype=torch.float32, device=dev)))
names = ["top/flush", "setup_ray", "first try", "pool pop", "retry search", "advance", "trace", "finish+store", "push", "tail"]
rows = []
for i, s in enumerate(samples):
    for j, o in enumerate(outs):
        for _ in range(3):
            cam.create_rays(s, ray_index_base=0, out=o)
        torch.cuda.synchronize()
        lib.zoic_debug_region_cycles((ctypes.c_ulonglong * 16)(), 1)      # reset
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        steps = 5
        for _ in range(steps):
            cam.create_rays(s, ray_index_base=0, out=o)
        e1.record()
        torch.cuda.synchronize()
        rc = (ctypes.c_ulonglong * 16)()
        lib.zoic_debug_region_cycles(rc, 1)
        rate = n * steps / (e0.elapsed_time(e1) * 1e-3) / 1e9
        rows.append((rate, i, j, [int(v) / steps for v in rc]))
rows.sort()
print("%s %s, stats build: Grays/s by pair, then wave-cycles per section in units of the FASTEST pair's total" % (name, prec))
ref = float(sum(rows[-1][3][:9]))
print("%-14s " % "pair" + " ".join("%12s" % x for x in names[:9]) + "        total")
for rate, i, j, rc in rows:
    print("s%d o%d %6.2f   " % (i, j, rate) + " ".join("%12.4f" % (rc[k] / ref) for k in range(9)) + "   %8.4f" % (sum(rc[:9]) / ref))
slow, fast = rows[0][3], rows[-1][3]
print("slowest - fastest, share of the difference: " + ", ".join("%s %.0f%%" % (names[k], 100.0 * (slow[k] - fast[k]) / max(sum(slow[:9]) - sum(fast[:9]), 1.0)) for k in range(9)))
cam.close()
