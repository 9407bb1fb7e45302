"""Round 6, second placement experiment: does plain streaming bandwidth depend on the allocation, or on the PAIR of allocations?
Four sample-sized and four ray-sized buffers; per buffer a read (sum) and a write (fill) rate, per pair a copy rate and C3's rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from zoic_amd.workloads import CONFIGS, ray_count

cfg = CONFIGS["C3"]
n = ray_count("C3")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


cam = bench.make_camera("C3", "fast", 0)
pads, samples, outs = [], [], []
s0 = cam.generate_samples(n, cfg["width"], cfg["height"], cfg["spp"], seed=1, ray_index_base=0)
for i in range(4):
    pads.append(torch.empty((7 + 11 * i) * 1024 * 1024 + 8192, dtype=torch.uint8, device=dev))
    samples.append(s0 if i == 0 else s0.clone())
    pads.append(torch.empty((5 + 3 * i) * 1024 * 1024 + 4096, dtype=torch.uint8, device=dev))
    outs.append(torch.empty((n, 8), dtype=torch.float32, device=dev))
sbytes = s0.numel() * s0.element_size()
print("sample buffers %d MB, ray buffers %d MB" % (sbytes >> 20, outs[0].numel() * 4 >> 20))
print("read  (sum) GB/s per sample buffer: " + " ".join("%.0f" % (sbytes / timed(lambda: s.view(torch.float32).sum()) / 1e9) for s in samples))
print("write (fill) GB/s per ray buffer:   " + " ".join("%.0f" % (o.numel() * 4 / timed(lambda: o.fill_(1.0)) / 1e9) for o in outs))
print("read  (sum) GB/s per ray buffer:    " + " ".join("%.0f" % (o.numel() * 4 / timed(lambda: o.sum()) / 1e9) for o in outs))
print("copy sample buffer (row) -> first half of ray buffer (column), GB/s read+write:")
for s in samples:
    sv = s.view(torch.float32).view(-1)
    print("  " + " ".join("%.0f" % (2 * sbytes / timed(lambda: o.view(-1)[:sv.numel()].copy_(sv)) / 1e9) for o in outs), flush=True)
print("C3 Grays/s, sample buffer (row) x ray buffer (column):")
for s in samples:
    row = []
    for o in outs:
        od = dict(rays=o)
        t = timed(lambda: cam.create_rays(s, ray_index_base=0, out=od), reps=30, warm=10)
        row.append("%.2f" % (n / t / 1e9))
    print("  " + " ".join(row), flush=True)
try:
    free, total = torch.cuda.mem_get_info()
    print("device memory: %.1f GB free of %.1f" % (free / 1e9, total / 1e9))
except Exception as e:
    print(e)
cam.close()
