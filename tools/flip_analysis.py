"""Where are FAST/STRICT decision flips decided?  Takes the flipped rays tools/flip_dump.py saved on the GPU box and asks
the oracle's margin probe (oracle/zoic_oracle.c: zo_create_rays_probe) which accept/reject decision of each ray's
evaluation was the closest call.    python tools/flip_analysis.py gpurun_out/r2a/flips"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from zoic_amd.workloads import camera_params, ray_rng_states

class Probe(C.Structure):
    _fields_ = [("m", C.c_float), ("iface", C.c_int), ("kind", C.c_int), ("tries", C.c_int)]

L = oracle.lib()
L.zo_create_rays_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
KIND = {0: "housing/stop clip", 1: "sphere miss", 2: "TIR"}
for cfg in ("C2", "C3", "C4", "C5"):
    f = os.path.join(sys.argv[1], "flips_%s.npz" % cfg)
    if not os.path.exists(f):
        continue
    d = np.load(f)
    idx, s, base = d["idx"], np.ascontiguousarray(d["samples"]), int(d["base"])
    oc = oracle.OracleCamera()
    oc.update(**camera_params(cfg)) if cfg != "C3" else None
    if cfg == "C3":
        from zoic_amd.workloads import hexagon_bokeh
        oc.set_bokeh_image(hexagon_bokeh()); oc.update(**camera_params(cfg))
    t = oc.lens_table()
    stop = t["apertureElement"]
    est = 5.96e-8 * abs(t["elements"][stop, 0]) / float(t["userApertureRadius"])
    print("%s: %d flipped rays; stop = interface %d, noise estimate eps*|R|/r_stop = %.2e" % (cfg, len(idx), stop, est))
    if not len(idx):
        continue
    st = np.ascontiguousarray(np.concatenate([ray_rng_states(1, 1, base + int(i)) for i in idx]))
    out = (Probe * len(idx))()
    L.zo_create_rays_probe(oc._h, len(idx), s.ctypes.data, st.ctypes.data, out)
    m = np.array([o.m for o in out]); ifc = np.array([o.iface for o in out]); kind = np.array([o.kind for o in out])
    for i in sorted(set(ifc)):
        for k in sorted(set(kind[ifc == i])):
            mm = m[(ifc == i) & (kind == k)]
            print("   closest call at interface %2d (%s): %5d rays, relative margin median %.2e  p90 %.2e  max %.2e" %
                  (i, KIND[k], len(mm), np.median(mm), np.percentile(mm, 90), mm.max()))
