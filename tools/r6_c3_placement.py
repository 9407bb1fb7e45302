"""Round 6: is C3's two-state rate (40.5-41 / 43.5-45.4 Grays/s) a matter of WHERE its buffers lie?  One process, several
cameras (each uploads its own 1.3 MB of cell records -> different physical pages), several sample / ray buffers, every
combination timed.  If the rate moves with the object, placement is the cause; if every combination of one process runs alike,
it is the process / box.  Usage (GPU box): python tools/r6_c3_placement.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from zoic_amd.workloads import CONFIGS, ray_count

cfg_name = "C3"
cfg = CONFIGS[cfg_name]
n = ray_count(cfg_name)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def rate(cam, samples, out, steps=30, warmup=10):
    for _ in range(warmup):
        cam.create_rays(samples, ray_index_base=0, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        cam.create_rays(samples, ray_index_base=0, out=out)
    e1.record()
    torch.cuda.synchronize()
    return n * steps / (e0.elapsed_time(e1) * 1e-3) / 1e9


cams, pads = [], []
for i in range(6):
    cams.append(bench.make_camera(cfg_name, "fast", 0))
    pads.append(torch.empty((3 + 5 * i) * 1024 * 1024 + 4096 * i, dtype=torch.uint8, device=dev))    # move the next camera's allocations along
samples = [cams[0].generate_samples(n, cfg["width"], cfg["height"], cfg["spp"], seed=1, ray_index_base=0)]
outs = [dict(rays=torch.empty((n, 8), dtype=torch.float32, device=dev))]
print("cameras (own tables each), one sample buffer, one ray buffer:")
for rep in range(2):
    print("  " + " ".join("%.2f" % rate(c, samples[0], outs[0]) for c in cams), flush=True)
for i in range(3):
    pads.append(torch.empty((7 + 11 * i) * 1024 * 1024 + 8192, dtype=torch.uint8, device=dev))
    samples.append(samples[0].clone())
    pads.append(torch.empty((5 + 3 * i) * 1024 * 1024 + 4096, dtype=torch.uint8, device=dev))
    outs.append(dict(rays=torch.empty((n, 8), dtype=torch.float32, device=dev)))
print("camera 0, sample buffer (row) x ray buffer (column):")
for s in samples:
    print("  " + " ".join("%.2f" % rate(cams[0], s, o) for o in outs), flush=True)
print("addresses: samples " + " ".join("%#x" % t.data_ptr() for t in samples) + " | rays " + " ".join("%#x" % o["rays"].data_ptr() for o in outs))
# the same bytes at other offsets of ONE allocation: is it the address (virtual offset) or the allocation?
big_o = torch.empty(6 * 1024 ** 3, dtype=torch.uint8, device=dev)
big_s = torch.empty(4 * 1024 ** 3, dtype=torch.uint8, device=dev)
print("big allocations: rays %#x samples %#x" % (big_o.data_ptr(), big_s.data_ptr()))
offs = [0, 4096, 65536, 524288, 2 << 20, 6 << 20, 32 << 20, 34 << 20, 256 << 20, 1 << 30]
row = []
for off in offs:
    o = dict(rays=big_o[off:off + n * 32].view(torch.float32).view(n, 8))
    row.append("%.2f" % rate(cams[0], samples[0], o))
print("ray buffer at offsets %s of one allocation (sample buffer 0): %s" % (offs, " ".join(row)), flush=True)
row = []
for off in offs:
    sv = big_s[off:off + samples[0].numel() * samples[0].element_size()].view(samples[0].dtype).view(samples[0].shape)
    sv.copy_(samples[0])
    row.append("%.2f" % rate(cams[0], sv, outs[0]))
print("sample buffer at those offsets of one allocation (ray buffer 0): %s" % " ".join(row), flush=True)
del big_o, big_s
print("camera 5, sample buffer (row) x ray buffer (column):")
for s in samples:
    print("  " + " ".join("%.2f" % rate(cams[5], s, o) for o in outs), flush=True)
for c in cams:
    c.close()
