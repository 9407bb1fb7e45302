"""One frame as 1 launch / 4 sub-launches on one stream / 4 sub-launches on alternating streams (ShardedFrame's layout).
    python tools/exp_sublaunch.py [--configs C4,C5] [--modes fast]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoic_amd import PRECISION_FAST, PRECISION_FAST_UNCHECKED, PRECISION_STRICT, ZoicCamera
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, ray_count
MODES = {'fast': PRECISION_FAST, 'unchecked': PRECISION_FAST_UNCHECKED, 'strict': PRECISION_STRICT}
ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="C4,C5")
ap.add_argument("--modes", default="fast,unchecked")
ap.add_argument("--parts", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda", 0)
for cfg in a.configs.split(","):
    c = CONFIGS[cfg]
    cam = ZoicCamera(0)
    if c["bokeh"]:
        cam.set_bokeh_image(hexagon_bokeh())
    cam.update(**camera_params(cfg))
    n = ray_count(cfg)
    s = cam.generate_samples(n, c["width"], c["height"], c["spp"], seed=1)
    out = torch.empty((n, 8), dtype=torch.float32, device=dev)
    cuts = [(n * i // a.parts) // 256 * 256 for i in range(a.parts)] + [n]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    def whole():
        cam.create_rays(s, out=dict(rays=out))
    def serial():
        for i in range(a.parts):
            lo, hi = cuts[i], cuts[i + 1]
            cam.create_rays(s[lo:hi], ray_index_base=lo, out=dict(rays=out[lo:hi]))
    def alternate():
        cur = torch.cuda.current_stream()
        for st in streams: st.wait_stream(cur)
        for i in range(a.parts):
            lo, hi = cuts[i], cuts[i + 1]
            with torch.cuda.stream(streams[i & 1]):
                cam.create_rays(s[lo:hi], ray_index_base=lo, out=dict(rays=out[lo:hi]))
        for st in streams: cur.wait_stream(st)
    for mode in a.modes.split(","):
        cam.set_precision(MODES[mode])
        for name, fn in (("whole", whole), ("serial", serial), ("alternate", alternate)):
            for _ in range(2): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 4
            print("%s %-9s %-9s %8.3f ms %7.2f Grays/s" % (cfg, mode, name, ms, n / ms / 1e6), flush=True)
    cam.close(); del s, out; torch.cuda.empty_cache()
