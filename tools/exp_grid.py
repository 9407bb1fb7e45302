"""tools/exp_grid.py [C2] [mode]: launch time against batch size for the persistent grid given by ZOIC_GRID_BLOCKS (one process per grid)."""
import os, subprocess, sys
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
mode = sys.argv[2] if len(sys.argv) > 2 else "unchecked"
if len(sys.argv) > 3:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from zoic_amd import ZoicCamera, PRECISION_FAST, PRECISION_FAST_UNCHECKED, PRECISION_STRICT
    from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh
    c = CONFIGS[cfg]
    cam = ZoicCamera(0)
    if c["bokeh"]:
        cam.set_bokeh_image(hexagon_bokeh())
    cam.update(**camera_params(cfg))
    cam.set_precision({"fast": PRECISION_FAST, "unchecked": PRECISION_FAST_UNCHECKED, "strict": PRECISION_STRICT}[mode])
    full = cam.generate_samples(c["width"] * c["height"] * 8, c["width"], c["height"], 8, seed=1)
    row = []
    for n in (1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 24):
        if n > full.shape[0]:
            break
        stride = full.shape[0] // n
        s = full[::stride][:n].contiguous()
        out = dict(rays=torch.empty((n, 8), dtype=torch.float32, device="cuda"))
        for _ in range(3):
            cam.create_rays(s, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            cam.create_rays(s, out=out)
        e1.record(); torch.cuda.synchronize()
        row.append("%7.1f" % (e0.elapsed_time(e1) / 20 * 1e3))
    print("grid %5s us: %s" % (os.environ.get("ZOIC_GRID_BLOCKS", "2048"), " ".join(row)), flush=True)
else:
    print("%s %s   n =    256K    512K      1M      2M      4M      8M     16M" % (cfg, mode))
    for g in (128, 256, 384, 512, 768, 1024, 1536, 2048):
        subprocess.run([sys.executable, __file__, cfg, mode, "child"], env=dict(os.environ, ZOIC_GRID_BLOCKS=str(g)))
