"""Rate of one config over every pair of (sample buffer, ray buffer) out of K allocations each, one process.
Usage: [ZOIC_AMD_LIB=...] python tools/r6_placement_matrix.py C3 fast 4"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from zoic_amd.workloads import CONFIGS, ray_count

name, prec, K = sys.argv[1], sys.argv[2], int(sys.argv[3])
cfg = CONFIGS[name]
n = ray_count(name)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cam = bench.make_camera(name, prec, 0)
pads, samples, outs = [], [], []
s0 = cam.generate_samples(n, cfg["width"], cfg["height"], cfg["spp"], seed=1, ray_index_base=0)
spacer = int(float(os.environ.get("SPACER_GB", "0")) * (1 << 30))
for i in range(K):
    if spacer and i > 0:
        pads.append(torch.empty(spacer, dtype=torch.uint8, device=dev))      # push the next candidates into another region of the device memory
    pads.append(torch.empty((7 + 11 * i) * 1024 * 1024 + 8192, dtype=torch.uint8, device=dev))
    samples.append(s0 if i == 0 else s0.clone())
    pads.append(torch.empty((5 + 3 * i) * 1024 * 1024 + 4096, dtype=torch.uint8, device=dev))
    outs.append(dict(rays=torch.empty((n, 8), dtype=torch.float32, device=dev)))
steps = max(4, min(30, int(0.1 / (n / 45e9))))
vals = []
for s in samples:
    row = []
    for o in outs:
        for _ in range(max(2, steps // 3)):
            cam.create_rays(s, ray_index_base=0, out=o)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            cam.create_rays(s, ray_index_base=0, out=o)
        e1.record()
        torch.cuda.synchronize()
        row.append(n * steps / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    vals += row
    print("  " + " ".join("%.2f" % v for v in row), flush=True)
print("addresses: samples " + " ".join("%#x" % t.data_ptr() for t in samples) + " | rays " + " ".join("%#x" % o["rays"].data_ptr() for o in outs))
print("%s %s spacer %s GB lib=%s: min %.2f mean %.2f max %.2f Grays/s over %d pairs" % (name, prec, os.environ.get("SPACER_GB", "0"), os.path.basename(os.environ.get("ZOIC_AMD_LIB", "default")), min(vals), sum(vals) / len(vals), max(vals), len(vals)))
cam.close()
