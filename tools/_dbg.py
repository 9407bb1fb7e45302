import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from zoic_amd import ZoicCamera, PRECISION_STRICT
from zoic_amd.workloads import *
cam = ZoicCamera(0); cam.update(**camera_params("C2")); cam.set_precision(PRECISION_STRICT)
c = CONFIGS["C2"]; base = int(c["width"]*int(c["height"]*0.03))*c["spp"]
s = synthetic_samples(4000, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base)
first = cam.create_rays(s)
k = int(np.argmax((first["tries"] > 0) & (first["weight"] != 0)))
print("k", k, "tries", first["tries"][k], "dir", first["dir"][:, k])
row = [float(v) for v in s[k]]
for tid in (5, 5, 5, 6, 0, 0):
    o = cam.create_ray(*row, tid=tid)
    print(tid, o.dir.x, o.dir.y, o.dir.z, o.weight[0], o.dOdy.x)
# explicit states
for seed in (1, 2, 3):
    st = ray_rng_states(1, seed=seed, ray_index_base=77)
    r = cam.create_rays(s[k:k+1], rng_states=st)
    print("states seed", seed, r["dir"][:, 0], r["tries"])
