"""Which listed rays differ between the long-list path and the short-list path, and which of the two agrees with the per-sample kernel?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoic_amd import ZoicCamera, PRECISION_FAST
from zoic_amd.workloads import CONFIGS, camera_params, ray_rng_states, synthetic_samples
cfg = "C4"
c = CONFIGS[cfg]
cam = ZoicCamera(0)
cam.update(**camera_params(cfg))
cam.set_precision(PRECISION_FAST)
n, base = 1 << 24, (c["width"] * (c["height"] // 3)) * c["spp"]
s = cam.generate_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base)
one = cam.create_rays(s, ray_index_base=base)["rays"].clone()
parts = torch.empty_like(one)
step = 1 << 20
for a in range(0, n, step):
    parts[a:a + step] = cam.create_rays(s[a:a + step], ray_index_base=base + a)["rays"]
torch.cuda.synchronize()
bad = (one.view(torch.int32) != parts.view(torch.int32)).any(1).nonzero().flatten().cpu().numpy()
A, Bp = one.cpu().numpy(), parts.cpu().numpy()
fa, fb = A[:, 7].view(np.uint32), Bp[:, 7].view(np.uint32)
print("differ:", len(bad), "flags differ:", int((fa[bad] != fb[bad]).sum()))
ta, tb = (fa[bad] >> 1) & 31, (fb[bad] >> 1) & 31
print("tries long :", np.bincount(ta, minlength=8)[:8])
print("tries short:", np.bincount(tb, minlength=8)[:8])
sh = s.cpu().numpy()
# per-ray streams given explicitly through rng_states: the per-sample kernel needs the tid stream instead; use the batch API with n=1 launches? -> short list of 1
for i in bad[:6]:
    r1 = cam.create_rays(s[i:i + 1].clone(), ray_index_base=base + int(i))["rays"].cpu().numpy()[0]
    print(int(i), "long", A[i, 3:7], ta[list(bad).index(i)], "short", Bp[i, 3:7], tb[list(bad).index(i)], "single", r1[3:7], (r1[7:8].view(np.uint32)[0] >> 1) & 31)
    print("    dlong-dshort", A[i, :7] - Bp[i, :7])
