"""Host-side mirror of zoic's Arnold camera node over the C-ABI (libzoic_amd.so).

    cam = ZoicCamera(device=0)                       # node_initialize   zoic.cpp:1565-1572
    cam.update(lensModel=RAYTRACED, lensDataPath=..) # node_update       zoic.cpp:1575-1720 (14 parameters, same names)
    rays = cam.create_rays(samples)                  # camera_create_ray zoic.cpp:1752-1990, batched
    cam.close()                                      # node_finish       zoic.cpp:1723-1749

Parameter names, defaults and error behaviour follow node_parameters (zoic.cpp:1547-1562).  Errors the reference
reports with AiMsgError + AiRenderAbort() surface as ZoicError with the same message text.

This module only marshals pointers; all rays come from the HIP kernels.  torch tensors (device memory) and numpy
arrays (host memory) are both accepted.
"""
import ctypes as C
import weakref
import os

import numpy as np

from . import _capi
from ._capi import PRECISION_FAST, PRECISION_FAST_UNCHECKED, PRECISION_STRICT, RAYTRACED, THINLENS  # noqa: F401

LENS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lenses")

# node_parameters, zoic.cpp:1547-1562
DEFAULTS = dict(sensorWidth=3.6, sensorHeight=2.4, focalLength=2.0, fStop=4.0, focalDistance=100.0, useImage=False,
                bokehPath="", lensModel=RAYTRACED, lensDataPath="", kolbSamplingLUT=True, useDof=True,
                opticalVignettingDistance=0.0, opticalVignettingRadius=1.0, exposureControl=0.0)
_STR = ("bokehPath", "lensDataPath")
_INT = ("useImage", "lensModel", "kolbSamplingLUT", "useDof")

FLAG_RETRIED = 1
FLAG_LUT_MISS = 64


def lens_path(name):
    """Path of a bundled lens prescription (zoic_amd/lenses/*.dat)."""
    p = os.path.join(LENS_DIR, name)
    if not os.path.exists(p):
        raise FileNotFoundError(p)
    return p


class ZoicError(RuntimeError):
    def __init__(self, status, detail):
        name = _capi.STATUS_NAMES[status] if 0 <= status < len(_capi.STATUS_NAMES) else str(status)
        super().__init__("%s: %s" % (name, detail))
        self.status = status
        self.status_name = name


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def rays_to_dict(rays):
    """(n,) zoic_ray records -> convenience views (planes are copies: ox oy oz dx dy dz weight)."""
    planes = np.stack([rays[k] for k in ("ox", "oy", "oz", "dx", "dy", "dz", "weight")]).astype(np.float32, copy=False)
    flags = rays["flags"].astype(np.uint8)
    return dict(rays=rays, planes=planes, origin=planes[0:3], dir=planes[3:6], weight=planes[6], flags=flags,
                tries=((flags >> 1) & 31).astype(np.int32))


class PinnedArray:
    """A numpy array over page-locked host memory (zoic_host_alloc): buffers of this kind let create_rays' host path
    run as an asynchronous two-stream pipeline.  Keep the object alive as long as the array is used."""

    def __init__(self, shape, dtype):
        self._lib = _capi.load()
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        p = C.c_void_p()
        st = self._lib.zoic_host_alloc(max(n, 1), C.byref(p))
        if st != 0:
            raise ZoicError(st, (self._lib.zoic_last_error_string() or b"").decode(errors="replace"))
        self._p = p
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if getattr(self, "_p", None):
            self.array = None
            self._lib.zoic_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ZoicTile:
    """A render thread's bucket of samples (zoic_tile_*): page-locked AtCameraInput / AtCameraOutput arrays the camera's
    RESIDENT kernel reads and writes in place -- no launch, no stream, no copy.

        tile = cam.tile(capacity=64 * 64 * 16, tid=3)
        tile.inputs[:n, (0, 1, 4, 5)] = samples       # (capacity, 7) float32 view: sx sy dsx dsy lensx lensy relative_time
        tile.submit(n, ray_index_base)                # returns at once
        tile.wait()                                   # tile.outputs[:n] -- (capacity, 21) float32 AtCameraOutput rows -- are complete
    Row i equals zoic_create_rays_arnold's row for the same sample and ray index, bit for bit."""

    def __init__(self, cam, capacity, tid=0):
        self._cam = cam
        self._lib = cam._lib
        h = C.c_void_p()
        self._h = None
        cam._check(self._lib.zoic_tile_create(cam._h, int(capacity), int(tid) & 0xFFFF, C.byref(h)))
        self._h = h
        cam._tiles.add(self)   # weak: ZoicCamera.close() closes the tiles still alive first (their views point into memory the camera's end frees)
        self.capacity = int(self._lib.zoic_tile_capacity(h))
        self.tid = int(tid)
        pin = C.cast(self._lib.zoic_tile_inputs(h), C.c_void_p).value
        pout = C.cast(self._lib.zoic_tile_outputs(h), C.c_void_p).value
        self.inputs = np.frombuffer((C.c_char * (self.capacity * 28)).from_address(pin), dtype=np.float32).reshape(self.capacity, 7)
        self.outputs = np.frombuffer((C.c_char * (self.capacity * 84)).from_address(pout), dtype=np.float32).reshape(self.capacity, 21)
        # the same memory as (capacity, 4) float32 samples (sx, sy, lensx, lensy): what the kernel reads after set_inputs(1)
        self.samples = np.frombuffer((C.c_char * (self.capacity * 16)).from_address(pin), dtype=np.float32).reshape(self.capacity, 4)
        # the same memory as (capacity, 8) float32 zoic_ray records: what the kernel writes after set_rows(1)
        self.rays = np.frombuffer((C.c_char * (self.capacity * 32)).from_address(pout), dtype=np.float32).reshape(self.capacity, 8)

    def set_rows(self, rows):
        """0: AtCameraOutput rows in `outputs` (the default); 1: zoic_ray records in `rays` (32 instead of 84 bytes a ray across PCIe)."""
        self._cam._check(self._lib.zoic_tile_set_rows(self._h, int(rows)))

    def set_inputs(self, inputs):
        """0: AtCameraInput rows in `inputs` (the default); 1: (sx, sy, lensx, lensy) in `samples` (16 instead of 28 bytes a sample)."""
        self._cam._check(self._lib.zoic_tile_set_inputs(self._h, int(inputs)))

    def submit(self, n, ray_index_base=0):
        self._cam._check(self._lib.zoic_tile_submit(self._h, int(n), int(ray_index_base)))

    def wait(self):
        self._cam._check(self._lib.zoic_tile_wait(self._h))

    def done(self):
        return bool(self._lib.zoic_tile_done(self._h))

    def close(self):
        if getattr(self, "_h", None):
            # every view onto the page-locked arrays goes before the arrays do
            self.inputs = self.outputs = self.samples = self.rays = None
            self._lib.zoic_tile_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ZoicCamera:
    def __init__(self, device=0):
        self._lib = _capi.load()
        h = C.c_void_p()
        self._h = None
        self._check(self._lib.zoic_camera_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.params = None
        self._tiles = weakref.WeakSet()   # the ZoicTile objects made by tile(): closed with the camera

    # ------------------------------------------------------------------ lifetime
    def _check(self, status):
        if status != 0:
            raise ZoicError(status, (self._lib.zoic_last_error_string() or b"").decode(errors="replace"))

    def close(self):
        if getattr(self, "_h", None):
            for t in list(getattr(self, "_tiles", ())):
                t.close()
            self._lib.zoic_camera_destroy(self._h)   # (would settle and detach them itself: the C-ABI's own rule, include/zoic_amd.h)
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

    # ------------------------------------------------------------------ node_update
    def set_bokeh_image(self, pixels):
        """pixels: (H, W, C>=3) float32 -- what AiTextureLoad would return for bokehPath (zoic.cpp:176-186)."""
        px = np.ascontiguousarray(pixels, dtype=np.float32)
        if px.ndim != 3:
            raise ValueError("bokeh image must be (H, W, C)")
        h, w, c = px.shape
        self._check(self._lib.zoic_camera_set_bokeh_image(self._h, w, h, c, px.ctypes.data))

    def set_lens_text(self, text):
        if text is None:
            self._check(self._lib.zoic_camera_set_lens_text(self._h, None, 0))
            return
        b = text.encode() if isinstance(text, str) else bytes(text)
        self._check(self._lib.zoic_camera_set_lens_text(self._h, b, len(b)))

    def set_precision(self, mode):
        self._check(self._lib.zoic_camera_set_precision(self._h, int(mode)))

    def set_wait_mode(self, mode):
        """0 spin (default), 1 yield, 2 sleep: how a calling thread waits for the resident kernel (zoic_camera_set_wait_mode)."""
        self._check(self._lib.zoic_camera_set_wait_mode(self._h, int(mode)))

    def set_seed(self, seed):
        self._check(self._lib.zoic_camera_set_seed(self._h, int(seed) & 0xFFFFFFFF))

    def set_frame_aspect(self, max_abs_sy):
        """The largest |sy| the renderer sends (1/aspect): the extent node_update's self-check of the FAST modes probes."""
        self._check(self._lib.zoic_camera_set_frame_aspect(self._h, float(max_abs_sy)))

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
        self._check(self._lib.zoic_camera_update(self._h, C.byref(P)))
        return self

    # ------------------------------------------------------------------ camera_create_ray
    def create_rays(self, samples, rng_states=None, ray_index_base=0, out=None, stream=None):
        """samples: (n,4) float32 rows (sx, sy, lensx, lensy).

        numpy in  -> host API (H2D, kernels, D2H); returns a dict of numpy arrays built from the (n,) zoic_ray records:
                     rays (structured), planes (7,n) = ox oy oz dx dy dz weight, origin (3,n), dir (3,n), weight, flags, tries.
        torch device tensor in -> device API, asynchronous on `stream` (default: torch's current stream); returns a dict
                     with rays = (n,8) float32 tensor (column 7 holds the flag word's bits) and strided views into it.
        """
        if _is_torch(samples):
            return self._create_rays_torch(samples, rng_states, ray_index_base, out, stream)
        s = np.ascontiguousarray(samples, dtype=np.float32)
        if s.ndim != 2 or s.shape[1] != 4:
            raise ValueError("samples must be (n, 4)")
        n = s.shape[0]
        if out is not None:   # caller-owned (e.g. pinned) record array
            rays = out
            if rays.dtype != np.dtype(_capi.RAY_DTYPE) or rays.shape != (n,) or not rays.flags.c_contiguous:
                raise ValueError("out must be a contiguous (n,) array of zoic_ray records")
        else:
            rays = np.empty(n, dtype=_capi.RAY_DTYPE)
        rs_ptr = None
        if rng_states is not None:
            rs = np.ascontiguousarray(rng_states, dtype=np.uint32)
            if rs.shape != (n, 4):
                raise ValueError("rng_states must be (n, 4) uint32")
            rs_ptr = rs.ctypes.data
        self._check(self._lib.zoic_create_rays_host(self._h, n, s.ctypes.data, rs_ptr, int(ray_index_base), rays.ctypes.data))
        return rays_to_dict(rays)

    def _create_rays_torch(self, samples, rng_states, ray_index_base, out, stream):
        import torch
        if samples.dtype != torch.float32 or samples.dim() != 2 or samples.shape[1] != 4 or not samples.is_contiguous():
            raise ValueError("samples must be a contiguous (n,4) float32 tensor")
        if not samples.is_cuda:
            raise ValueError("torch samples must live on the GPU (use numpy for host buffers)")
        if samples.device.index != self.device:
            raise ValueError("samples live on cuda:%s but this camera is bound to device %d" % (samples.device.index, self.device))
        n = samples.shape[0]
        if out is None:
            out = dict(rays=torch.empty((n, 8), dtype=torch.float32, device=samples.device))
        rays = out["rays"]
        if rays.device != samples.device:
            raise ValueError("out['rays'] must live on the samples' device")
        rs_ptr = None
        if rng_states is not None:
            if rng_states.dtype not in (torch.int32, torch.uint32) or tuple(rng_states.shape) != (n, 4):
                raise ValueError("rng_states must be (n,4) int32/uint32 on the device")
            rs_ptr = rng_states.data_ptr()
        st = stream if stream is not None else torch.cuda.current_stream(samples.device).cuda_stream
        self._check(self._lib.zoic_create_rays_device(self._h, n, samples.data_ptr(), rs_ptr, int(ray_index_base),
                                                      rays.data_ptr(), C.c_void_p(st)))
        out.update(origin=rays[:, 0:3].t(), dir=rays[:, 3:6].t(), weight=rays[:, 6], planes=rays[:, 0:7].t(),
                   flags=rays[:, 7].view(torch.int32))
        return out

    def create_rays_device_ptr(self, n, d_samples, d_rays, d_rng=None, ray_index_base=0, stream=0):
        """Raw-pointer form of zoic_create_rays_device (d_rays: n x 32-byte zoic_ray records)."""
        self._check(self._lib.zoic_create_rays_device(self._h, n, d_samples, d_rng, int(ray_index_base), d_rays,
                                                      C.c_void_p(stream)))

    def create_rays_resident(self, samples, ray_index_base=0, out=None, tid=0):
        """(n, 4) float32 samples on the camera's device -> (n, 8) float32 zoic_ray records on the device through the RESIDENT kernel
        (zoic_create_rays_device_resident): no launch, not stream-ordered -- the current stream is synchronised first (the samples must
        be complete), the records are complete on return.  The bits of create_rays(samples, ray_index_base=...)."""
        import torch
        if samples.dtype != torch.float32 or samples.dim() != 2 or samples.shape[1] != 4 or not samples.is_contiguous():
            raise ValueError("samples must be a contiguous (n, 4) float32 device tensor")
        n = samples.shape[0]
        if out is None:
            out = torch.empty((n, 8), dtype=torch.float32, device=samples.device)
        if tuple(out.shape) != (n, 8) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != samples.device:
            raise ValueError("out must be a contiguous (n, 8) float32 tensor on the samples' device")
        torch.cuda.current_stream(samples.device).synchronize()
        self._check(self._lib.zoic_create_rays_device_resident(self._h, n, samples.data_ptr(), out.data_ptr(), int(ray_index_base), int(tid) & 0xFFFF))
        return out

    def create_ray(self, sx, sy, lensx, lensy, tid=0):
        """The per-sample camera_create_ray(node, input, output, tid) signature (one AtCameraInput in, one AtCameraOutput out)."""
        i = _capi.CameraInput(sx, sy, 0.0, 0.0, lensx, lensy, 0.0)
        o = _capi.CameraOutput()
        o.weight[0] = o.weight[1] = o.weight[2] = 1.0
        self._check(self._lib.zoic_camera_create_ray(self._h, C.byref(i), C.byref(o), int(tid)))
        return o

    def reverse_ray(self, Po=(0.0, 0.0, 0.0), fov=0.0):
        """camera_reverse_ray (zoic.cpp:1992-1995): always False, nothing written."""
        po = _capi.Vec3(*[float(v) for v in Po])
        ps = (C.c_float * 2)(0.0, 0.0)
        t = C.c_float(0.0)
        return bool(self._lib.zoic_camera_reverse_ray(self._h, C.byref(po), float(fov), ps, C.byref(t)))

    def create_rays_arnold(self, inputs, ray_index_base=0):
        """inputs: (n,7) float32 AtCameraInput rows -> (n,21) float32 AtCameraOutput rows (weight initialised to 1)."""
        a = np.ascontiguousarray(inputs, dtype=np.float32)
        n = a.shape[0]
        outs = np.zeros((n, 21), dtype=np.float32)
        outs[:, 18:21] = 1.0
        self._check(self._lib.zoic_create_rays_arnold(self._h, n, a.ctypes.data_as(C.POINTER(_capi.CameraInput)),
                                                      outs.ctypes.data_as(C.POINTER(_capi.CameraOutput)), int(ray_index_base)))
        return outs

    def tile(self, capacity, tid=0):
        """A ZoicTile of this camera (closed with the camera at the latest: ZoicCamera.close() closes the tiles still alive)."""
        return ZoicTile(self, capacity, tid)

    def create_rays_tile(self, inputs, ray_index_base=0, tid=0, out=None):
        """(n,7) float32 AtCameraInput rows -> (n,21) float32 AtCameraOutput rows through the resident kernel
        (zoic_camera_create_rays_tile): page-locked arrays (PinnedArray) are used in place, numpy arrays are staged."""
        a = inputs if (isinstance(inputs, np.ndarray) and inputs.dtype == np.float32 and inputs.flags.c_contiguous) else np.ascontiguousarray(inputs, dtype=np.float32)
        n = a.shape[0]
        outs = out if out is not None else np.empty((n, 21), dtype=np.float32)
        self._check(self._lib.zoic_camera_create_rays_tile(self._h, n, a.ctypes.data_as(C.POINTER(_capi.CameraInput)),
                                                           outs.ctypes.data_as(C.POINTER(_capi.CameraOutput)), int(ray_index_base), int(tid) & 0xFFFF))
        return outs

    def generate_samples(self, n, width, height, spp, seed=1, ray_index_base=0, out=None, stream=None):
        """Synthetic camera samples on the device (torch tensor out)."""
        import torch
        dev = torch.device("cuda", self.device)
        if out is None:
            out = torch.empty((n, 4), dtype=torch.float32, device=dev)
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        self._check(self._lib.zoic_generate_samples_device(self._h, n, int(ray_index_base), width, height, spp, seed,
                                                           out.data_ptr(), C.c_void_p(st)))
        return out

    # ------------------------------------------------------------------ statistics / tables
    def counters(self):
        c = _capi.Counters()
        self._check(self._lib.zoic_camera_get_counters(self._h, C.byref(c)))
        return dict(succesRays=c.succesRays, vignettedRays=c.vignettedRays, totalInternalReflection=c.totalInternalReflection)

    def reset_counters(self):
        self._check(self._lib.zoic_camera_reset_counters(self._h))

    def info(self):
        i = _capi.LensInfo()
        self._check(self._lib.zoic_camera_get_info(self._h, C.byref(i)))
        n = i.lensCount
        f = lambda a, m: np.array(a[:m], dtype=np.float32)  # noqa: E731
        elements = np.stack([f(i.curvature, n), f(i.thickness, n), f(i.ior, n), f(i.aperture, n), f(i.center, n)], 1) \
            if n else np.zeros((0, 5), np.float32)
        return dict(lensCount=n, apertureElement=i.apertureElement, elements=elements,
                    userApertureRadius=np.float32(i.userApertureRadius), originShift=np.float32(i.originShift),
                    apertureDistance=np.float32(i.apertureDistance), focalLengthRatio=np.float32(i.focalLengthRatio),
                    tracedFocalLength=(np.float32(i.tracedFocalLength[0]), np.float32(i.tracedFocalLength[1])),
                    fov=np.float32(i.fov), tan_fov=np.float32(i.tan_fov), apertureRadius=np.float32(i.apertureRadius),
                    lutKeys=f(i.lutKey, i.lutSize),
                    lutBoxes=np.stack([f(i.lutMaxX, i.lutSize), f(i.lutMaxY, i.lutSize), f(i.lutMinX, i.lutSize),
                                       f(i.lutMinY, i.lutSize)], 1),
                    bokehWidth=i.bokehWidth, bokehHeight=i.bokehHeight, fastRunsStrict=bool(i.fastRunsStrict), precomputeTIR=int(i.precomputeTIR))

    def bokeh_tables(self):
        i = self.info()
        x, y = i["bokehWidth"], i["bokehHeight"]
        if x <= 0 or y <= 0:
            return None
        t = dict(x=x, y=y, cdfRow=np.empty(y, np.float32), rowIndices=np.empty(y, np.int32),
                 cdfColumn=np.empty(x * y, np.float32), columnIndices=np.empty(x * y, np.int32))
        self._check(self._lib.zoic_camera_get_bokeh_tables(self._h, t["cdfRow"].ctypes.data, t["rowIndices"].ctypes.data,
                                                           t["cdfColumn"].ctypes.data, t["columnIndices"].ctypes.data))
        return t
