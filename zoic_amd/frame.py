"""Host-side mirror of zoic_frame_* (include/zoic_amd.h): ONE camera node spread over several HIP devices of this process.

    frame = ZoicFrame(devices=[0, 1, 2, 3])            # node_initialize on every device; devices[0] is the root
    frame.update(lensModel=RAYTRACED, lensDataPath=..)  # node_update on every device (identical tables)
    frame.generate_samples(n, W, H, spp)                # every device its own ray-index slab
    rays = frame.render(n)                              # (n, 8) records on the root device, global ray order

This is the reference's process model (one process, zoic.cpp:1752 called from every render thread); the
one-process-per-GPU form over torch.distributed lives in sharding.py.  Only pointers are marshalled here.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import FRAME_PAYLOAD, FRAME_RECORDS  # noqa: F401
from .camera import DEFAULTS, _INT, _STR, ZoicError


def frame_slab(n, n_devices, i):
    """[begin, end) of device i's slab (zoic_frame_slab: pure arithmetic in the library, no device needed)."""
    lib = _capi.load()
    a, b = C.c_uint64(), C.c_uint64()
    st = lib.zoic_frame_slab(int(n), int(n_devices), int(i), C.byref(a), C.byref(b))
    if st != 0:
        raise ZoicError(st, (lib.zoic_last_error_string() or b"").decode(errors="replace"))
    return a.value, b.value


class ZoicFrame:
    def __init__(self, devices=(0,)):
        self._lib = _capi.load()
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        self._h = None
        self._check(self._lib.zoic_frame_create(arr, len(self.devices), C.byref(h)))
        self._h = h
        self.params = None

    def _check(self, status):
        if status != 0:
            raise ZoicError(status, (self._lib.zoic_last_error_string() or b"").decode(errors="replace"))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.zoic_frame_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ------------------------------------------------------------------ node_update on every device
    def set_bokeh_image(self, pixels):
        px = np.ascontiguousarray(pixels, dtype=np.float32)
        h, w, c = px.shape
        self._check(self._lib.zoic_frame_set_bokeh_image(self._h, w, h, c, px.ctypes.data))

    def set_lens_text(self, text):
        b = None if text is None else (text.encode() if isinstance(text, str) else bytes(text))
        self._check(self._lib.zoic_frame_set_lens_text(self._h, b, len(b) if b else 0))

    def set_precision(self, mode):
        self._check(self._lib.zoic_frame_set_precision(self._h, int(mode)))

    def set_seed(self, seed):
        self._check(self._lib.zoic_frame_set_seed(self._h, int(seed) & 0xFFFFFFFF))

    def set_chunk_rays(self, rays):
        self._check(self._lib.zoic_frame_set_chunk_rays(self._h, int(rays)))

    def update(self, **kw):
        unknown = set(kw) - set(DEFAULTS)
        if unknown:
            raise KeyError("unknown zoic parameter(s): %s" % sorted(unknown))
        p = dict(DEFAULTS)
        p.update(kw)
        P = _capi.Params()
        self._keep = []
        for k, v in p.items():
            if k in _STR:
                b = str(v).encode()
                self._keep.append(b)
                setattr(P, k, b)
            elif k in _INT:
                setattr(P, k, int(v))
            else:
                setattr(P, k, float(v))
        self.params = p
        self._check(self._lib.zoic_frame_update(self._h, C.byref(P)))
        return self

    # ------------------------------------------------------------------ camera_create_ray over the devices
    def slab(self, n, i):
        return frame_slab(n, len(self.devices), i)

    def generate_samples(self, n, width, height, spp, seed=1, ray_index_base=0):
        self._check(self._lib.zoic_frame_generate_samples(self._h, int(n), int(ray_index_base), width, height, spp, seed))

    def _sample_ptrs(self, samples):
        if samples is None:
            return None
        if len(samples) != len(self.devices):
            raise ValueError("one samples tensor per device")
        return (C.c_void_p * len(samples))(*[(t.data_ptr() if t is not None and t.numel() else None) for t in samples])

    def render(self, n, samples=None, ray_index_base=0, out=None, layout=FRAME_RECORDS, stream=None):
        """samples: None (generate_samples' slabs) or one (slab, 4) float32 tensor per device.  Returns the root's
        (n, 8) records tensor (column 7 = the flag word's bits) or (n, 7) payload tensor; asynchronous on `stream`
        (default: torch's current stream of the root device)."""
        import torch
        dev = torch.device("cuda", self.devices[0])
        cols = 8 if layout == FRAME_RECORDS else 7
        if out is None:
            out = torch.empty((n, cols), dtype=torch.float32, device=dev)
        if tuple(out.shape) != (n, cols) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
            raise ValueError("out must be a contiguous (n, %d) float32 tensor on the root device" % cols)
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        self._check(self._lib.zoic_frame_render_device(self._h, int(n), self._sample_ptrs(samples), int(ray_index_base),
                                                       out.data_ptr(), int(layout), C.c_void_p(st)))
        return out

    def render_local(self, n, samples=None, ray_index_base=0):
        """Every device renders its slab into the frame's own buffers; nothing moves (compute-only leg)."""
        self._check(self._lib.zoic_frame_render_local(self._h, int(n), self._sample_ptrs(samples), int(ray_index_base), None))

    def render_host(self, samples, ray_index_base=0, out=None):
        """numpy (n, 4) samples -> (n,) zoic_ray records; every device moves its slab over its own PCIe link."""
        s = np.ascontiguousarray(samples, dtype=np.float32)
        n = s.shape[0]
        rays = out if out is not None else np.empty(n, dtype=_capi.RAY_DTYPE)
        self._check(self._lib.zoic_frame_render_host(self._h, n, s.ctypes.data, int(ray_index_base), rays.ctypes.data))
        return rays

    def synchronize(self):
        self._check(self._lib.zoic_frame_synchronize(self._h))

    def lane_info(self, i):
        """What device i's lane did in the last render() and how it reaches the root (zoic_frame_get_lane_info)."""
        info = _capi.FrameLaneInfo()
        self._check(self._lib.zoic_frame_get_lane_info(self._h, int(i), C.byref(info)))
        return dict(device=info.device, peer_access_to_root=bool(info.peer_access_to_root), peer_access_from_root=bool(info.peer_access_from_root),
                    chunks=info.chunks, rays=info.rays, bytes_to_root=info.bytes_to_root)

    def auto_layout(self):
        """(layout FRAME_PAYLOAD_AUTO has chosen for the tables the frame holds: FRAME_PAYLOAD / FRAME_PAYLOAD_SPARSE, None while undecided;
        the zero-weight fraction it measured, None while undecided) -- zoic_frame_auto_layout."""
        z = C.c_double(-1.0)
        r = self._lib.zoic_frame_auto_layout(self._h, C.byref(z))
        return (None if r < 0 else int(r)), (None if z.value < 0 else z.value)

    def counters(self):
        c = _capi.Counters()
        self._check(self._lib.zoic_frame_get_counters(self._h, C.byref(c)))
        return dict(succesRays=c.succesRays, vignettedRays=c.vignettedRays, totalInternalReflection=c.totalInternalReflection)
