"""C3's rate against WHERE in one large allocation its two buffers lie (GB-scale offsets).  If the allocation is physically
contiguous the pattern shows which address bits relate the sample stream and the ray stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from zoic_amd.workloads import CONFIGS, ray_count

name = "C3"
cfg = CONFIGS[name]
n = ray_count(name)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cam = bench.make_camera(name, "fast", 0)
s0 = cam.generate_samples(n, cfg["width"], cfg["height"], cfg["spp"], seed=1, ray_index_base=0)
GB = 1 << 30
total = int(os.environ.get("SCAN_GB", "240"))
big = torch.empty(total * GB, dtype=torch.uint8, device=dev)
print("allocation of %d GB at %#x" % (total, big.data_ptr()))
sb, ob = s0.numel() * s0.element_size(), n * 32


def view_s(off):
    v = big[off:off + sb].view(s0.dtype).view(s0.shape)
    v.copy_(s0)
    return v


def view_o(off):
    return dict(rays=big[off:off + ob].view(torch.float32).view(n, 8))


def rate(s, o, steps=20, warm=6):
    for _ in range(warm):
        cam.create_rays(s, ray_index_base=0, out=o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        cam.create_rays(s, ray_index_base=0, out=o)
    e1.record()
    torch.cuda.synchronize()
    return n * steps / (e0.elapsed_time(e1) * 1e-3) / 1e9


step = int(os.environ.get("SCAN_STEP_GB", "4"))
s = view_s(0)
print("samples at 0; ray buffer at k GB, k = 4 .. %d step %d:" % (total - 4, step))
print(" ".join("%d:%.1f" % (k, rate(s, view_o(k * GB))) for k in range(4, total - 4, step)), flush=True)
o = view_o(0)
print("ray buffer at 0; samples at k GB:")
print(" ".join("%d:%.1f" % (k, rate(view_s(k * GB), o)) for k in range(4, total - 2, step)), flush=True)
mid = (total // 2) // step * step
s = view_s(mid * GB)
print("samples at %d GB; ray buffer at k GB:" % mid)
print(" ".join("%d:%.1f" % (k, rate(s, view_o(k * GB))) for k in list(range(0, mid - 4, step)) + list(range(mid + 4, total - 4, step))), flush=True)
cam.close()
