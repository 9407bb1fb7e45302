"""Workload definitions shared by bench.py and the tests: BASELINE.json's five configs with the camera
parameters SURVEY.md 8(d) pins, the synthetic sample lattice, the per-ray retry-stream seeding and the
procedural bokeh image.  numpy only -- the device generators in csrc/kernels.hip and csrc/optics.hpp produce the
same numbers bit for bit (tests/test_samples.py, tests/test_parity_gpu.py).
"""
import numpy as np

from .camera import RAYTRACED, THINLENS, lens_path

_COMMON = dict(sensorWidth=3.6, sensorHeight=2.4, focalDistance=100.0, useDof=True, kolbSamplingLUT=True,
               exposureControl=0.0, opticalVignettingRadius=1.0)

# name -> (width, height, spp, camera parameters, needs bokeh image)
CONFIGS = {
    "C1": dict(width=1920, height=1080, spp=4, bokeh=False,
               desc="Thin-lens f/2.8, 1920x1080x4spp, no bokeh image",
               params=dict(_COMMON, lensModel=THINLENS, focalLength=5.0, fStop=2.8, opticalVignettingDistance=0.0, useImage=False)),
    "C2": dict(width=1920, height=1080, spp=8, bokeh=False,
               desc="Kolb raytraced F_2.8_TESSAR, 1920x1080x8spp",
               params=dict(_COMMON, lensModel=RAYTRACED, lens="tessar_f2.8.dat", focalLength=10.0, fStop=2.8, useImage=False)),
    "C3": dict(width=3840, height=2160, spp=16, bokeh=True,
               desc="F_2.0_DOUBLE_GAUSS + image-based bokeh CDF sampler, 3840x2160x16spp",
               params=dict(_COMMON, lensModel=RAYTRACED, lens="double_gauss_f2.0.dat", focalLength=5.0, fStop=2.0, useImage=True,
                           bokehPath="procedural:hexagon256")),
    "C4": dict(width=3840, height=2160, spp=32, bokeh=False,
               desc="F_4.0_FISHEYE_MULLER, 3840x2160x32spp",
               params=dict(_COMMON, lensModel=RAYTRACED, lens="fisheye_muller_f4.0.dat", focalLength=1.6, fStop=4.0, useImage=False)),
    "C5": dict(width=7680, height=4320, spp=64, bokeh=False,
               desc="F_1.25_PETZVAL wide open, 7680x4320x64spp (reject-rate stress)",
               params=dict(_COMMON, lensModel=RAYTRACED, lens="petzval_f1.25.dat", focalLength=5.0, fStop=1.25, useImage=False)),
}


def camera_params(name):
    """Keyword arguments for ZoicCamera.update / OracleCamera.update for config `name`."""
    p = dict(CONFIGS[name]["params"])
    lens = p.pop("lens", None)
    if lens:
        p["lensDataPath"] = lens_path(lens)
    return p


def ray_count(name):
    c = CONFIGS[name]
    return c["width"] * c["height"] * c["spp"]


# ---------------------------------------------------------------------------------------------- hashing
def pcg_hash(v):
    """32-bit PCG output hash (vectorised); mirrors zoic::pcg_hash in csrc/optics.hpp."""
    v = np.asarray(v, dtype=np.uint32)
    with np.errstate(over="ignore"):
        state = v * np.uint32(747796405) + np.uint32(2891336453)
        word = ((state >> ((state >> np.uint32(28)) + np.uint32(4))) ^ state) * np.uint32(277803737)
    return (word >> np.uint32(22)) ^ word


def _u01_24(h):
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(5.9604644775390625e-08)


def synthetic_samples(n, width, height, spp, seed=1, ray_index_base=0):
    """(n,4) float32 (sx, sy, lensx, lensy): pixel-jittered screen samples + uniform lens samples.

    ray id = (py*W + px)*spp + s; mirrors generate_samples_kernel (csrc/kernels.hip)."""
    ids = np.arange(ray_index_base, ray_index_base + n, dtype=np.uint64)
    pix = ids // np.uint64(spp)
    px = (pix % np.uint64(width)).astype(np.uint32)
    py = ((pix // np.uint64(width)) % np.uint64(height)).astype(np.uint32)
    lo = (ids & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (ids >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        key = np.uint32(seed) ^ pcg_hash(hi ^ np.uint32(0x632BE5AB))
        jx = _u01_24(pcg_hash(key ^ (lo * np.uint32(4) + np.uint32(0))))
        jy = _u01_24(pcg_hash(key ^ (lo * np.uint32(4) + np.uint32(1)) ^ np.uint32(0x85EBCA6B)))
        lx = _u01_24(pcg_hash(key ^ (lo * np.uint32(4) + np.uint32(2)) ^ np.uint32(0xC2B2AE35)))
        ly = _u01_24(pcg_hash(key ^ (lo * np.uint32(4) + np.uint32(3)) ^ np.uint32(0x27D4EB2F)))
    fW, fH = np.float32(width), np.float32(height)
    aspect = fW / fH
    sx = np.float32(2.0) * (px.astype(np.float32) + jx) / fW - np.float32(1.0)
    sy = (np.float32(1.0) - np.float32(2.0) * (py.astype(np.float32) + jy) / fH) / aspect
    return np.stack([sx, sy, lx, ly], axis=1).astype(np.float32)


def ray_rng_states(n, seed=1, ray_index_base=0):
    """(n,4) uint32 xorshift128 states of the per-ray retry streams; mirrors zoic::rng_for_ray (csrc/optics.hpp)."""
    ids = np.arange(ray_index_base, ray_index_base + n, dtype=np.uint64)
    lo = (ids & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (ids >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        k = pcg_hash(np.uint32(seed) ^ pcg_hash(hi + np.uint32(0x9E3779B9)))
        x = pcg_hash(k ^ (lo * np.uint32(4) + np.uint32(0)))
        y = pcg_hash(k ^ (lo * np.uint32(4) + np.uint32(1)) ^ np.uint32(0x85EBCA6B))
        z = pcg_hash(k ^ (lo * np.uint32(4) + np.uint32(2)) ^ np.uint32(0xC2B2AE35))
        w = pcg_hash(k ^ (lo * np.uint32(4) + np.uint32(3)) ^ np.uint32(0x27D4EB2F)) | np.uint32(1)
    return np.stack([x, y, z, w], axis=1).astype(np.uint32)


def hexagon_bokeh(size=256, seed=7):
    """Procedural (size,size,3) bokeh image of SURVEY 8(d): hexagon of circum-radius 0.9, value 0.3+0.7r inside,
    0 outside, plus 1e-3*u01(hash(pixel)) inside so the CDF sorts have no ties."""
    ii, jj = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    y = (ii.astype(np.float32) + np.float32(0.5)) / np.float32(size) * np.float32(2) - np.float32(1)
    x = (jj.astype(np.float32) + np.float32(0.5)) / np.float32(size) * np.float32(2) - np.float32(1)
    R = np.float32(0.9)
    k = np.float32(0.8660254)
    ax, ay = np.abs(x), np.abs(y)
    inside = (ay <= R * k) & (k * ax + np.float32(0.5) * ay <= R * k)
    r = np.sqrt(x * x + y * y)
    h = pcg_hash((ii * size + jj).astype(np.uint32) ^ np.uint32(seed))
    val = np.float32(0.3) + np.float32(0.7) * r + np.float32(1e-3) * _u01_24(h)
    lum = np.where(inside, val, np.float32(0)).astype(np.float32)
    return np.repeat(lum[:, :, None], 3, axis=2).copy()
