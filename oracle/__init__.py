"""ctypes binding of oracle/libzoic_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It is the CPU checker (a plain-C restatement of zoic.cpp's hot path, see
oracle/zoic_oracle.c) -- never part of the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZOIC_ORACLE_LIB: another build of the SAME source (tests/test_oracle_assumptions.py: other compilers / optimisation levels /
# the two variant hooks); never consulted by the product
_LIB_PATH = os.environ.get("ZOIC_ORACLE_LIB") or os.path.join(_HERE, "libzoic_oracle.so")

THINLENS, RAYTRACED, NONE = 0, 1, 2

ERR_NAMES = {0: "OK", 1: "LENS_PATH", 2: "LENS_COLUMNS", 3: "MULTI_APERTURE", 4: "NO_APERTURE",
             5: "BOKEH", 6: "LENS_PARSE", 7: "TOO_MANY_LENSES"}


def build(force=False):
    """Compile the C restatement with gcc (strict IEEE flags live in oracle/Makefile)."""
    src = os.path.join(_HERE, "zoic_oracle.c")
    if os.environ.get("ZOIC_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libzoic_oracle.so"])
    return _LIB_PATH


class Params(C.Structure):
    _fields_ = [("sensorWidth", C.c_float), ("sensorHeight", C.c_float), ("focalLength", C.c_float),
                ("fStop", C.c_float), ("focalDistance", C.c_float), ("useImage", C.c_int),
                ("lensModel", C.c_int), ("kolbSamplingLUT", C.c_int), ("useDof", C.c_int),
                ("opticalVignettingDistance", C.c_float), ("opticalVignettingRadius", C.c_float),
                ("exposureControl", C.c_float), ("bokehPath", C.c_char_p), ("lensDataPath", C.c_char_p)]


class LensElement(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("curvature", "thickness", "ior", "aperture", "abbe", "center")]


class V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class V2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Rng(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in "xyzw"]


# node_parameters defaults, zoic.cpp:1547-1562
DEFAULTS = dict(sensorWidth=3.6, sensorHeight=2.4, focalLength=2.0, fStop=4.0, focalDistance=100.0,
                useImage=False, bokehPath="", lensModel=RAYTRACED, lensDataPath="", kolbSamplingLUT=True,
                useDof=True, opticalVignettingDistance=0.0, opticalVignettingRadius=1.0, exposureControl=0.0)

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, f32p, u8p, u32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
        L.zo_camera_new.restype = vp
        L.zo_camera_free.argtypes = [vp]
        L.zo_camera_reset_rng.argtypes = [vp]
        L.zo_camera_rng.argtypes = [vp]; L.zo_camera_rng.restype = C.POINTER(Rng)
        L.zo_camera_set_bokeh_pixels.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
        L.zo_camera_set_lens_text.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.zo_camera_update.argtypes = [vp, C.POINTER(Params)]; L.zo_camera_update.restype = C.c_int
        L.zo_create_rays.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp]
        L.zo_create_rays_mt.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, C.c_int]
        for name, rt in [("zo_lens_count", C.c_int), ("zo_aperture_element", C.c_int),
                         ("zo_user_aperture_radius", C.c_float), ("zo_origin_shift", C.c_float),
                         ("zo_aperture_distance", C.c_float), ("zo_focal_length_ratio", C.c_float),
                         ("zo_lut_size", C.c_int), ("zo_fov", C.c_float), ("zo_tan_fov", C.c_float),
                         ("zo_aperture_radius", C.c_float)]:
            getattr(L, name).argtypes = [vp]; getattr(L, name).restype = rt
        L.zo_traced_focal_length.argtypes = [vp, C.c_int]; L.zo_traced_focal_length.restype = C.c_float
        L.zo_lenses.argtypes = [vp]; L.zo_lenses.restype = C.POINTER(LensElement)
        L.zo_lut_keys.argtypes = [vp]; L.zo_lut_keys.restype = f32p
        L.zo_lut_boxes.argtypes = [vp]; L.zo_lut_boxes.restype = f32p
        L.zo_surface_visits.argtypes = [vp]; L.zo_surface_visits.restype = C.c_longlong
        L.zo_counters.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.zo_bokeh_dims.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]; L.zo_bokeh_dims.restype = C.c_int
        L.zo_bokeh_cdf_row.argtypes = [vp]; L.zo_bokeh_cdf_row.restype = f32p
        L.zo_bokeh_cdf_column.argtypes = [vp]; L.zo_bokeh_cdf_column.restype = f32p
        L.zo_bokeh_row_indices.argtypes = [vp]; L.zo_bokeh_row_indices.restype = C.POINTER(C.c_int)
        L.zo_bokeh_column_indices.argtypes = [vp]; L.zo_bokeh_column_indices.restype = C.POINTER(C.c_int)
        L.zo_concentric_disk_sample.argtypes = [C.c_float, C.c_float, C.POINTER(V2)]
        L.zo_fast_sin.argtypes = [C.c_float]; L.zo_fast_sin.restype = C.c_float
        L.zo_fast_cos.argtypes = [C.c_float]; L.zo_fast_cos.restype = C.c_float
        L.zo_bokeh_sample.argtypes = [vp, C.c_float, C.c_float, f32p, f32p]
        L.zo_trace_record.argtypes = [vp, C.POINTER(V3), C.POINTER(V3), C.POINTER(V3), C.POINTER(C.c_int)]
        L.zo_trace_record.restype = C.c_int
        L.zo_xor128.argtypes = [C.POINTER(Rng)]; L.zo_xor128.restype = C.c_uint32
        L.zo_rng_seed.argtypes = [C.POINTER(Rng)]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__("oracle update failed: %s" % ERR_NAMES.get(code, code))
        self.code = code


class OracleCamera:
    """One zoic camera node evaluated by the CPU restatement (fresh xor128 state, like a fresh process)."""

    def __init__(self):
        self._L = lib()
        self._h = self._L.zo_camera_new()
        self.params = None

    def close(self):
        if self._h:
            self._L.zo_camera_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_bokeh_image(self, pixels):
        px = np.ascontiguousarray(pixels, dtype=np.float32)
        h, w, nc = px.shape
        self._L.zo_camera_set_bokeh_pixels(self._h, w, h, nc, px.ctypes.data)

    def set_lens_text(self, text):
        if isinstance(text, str):
            text = text.encode()
        self._L.zo_camera_set_lens_text(self._h, text, len(text))

    def update(self, **kw):
        p = dict(DEFAULTS); p.update(kw)
        unknown = set(p) - set(DEFAULTS)
        if unknown:
            raise KeyError("unknown zoic parameter(s): %s" % sorted(unknown))
        P = Params()
        for k, v in p.items():
            if k in ("bokehPath", "lensDataPath"):
                setattr(P, k, str(v).encode())
            elif k in ("useImage", "kolbSamplingLUT", "useDof", "lensModel"):
                setattr(P, k, int(v))
            else:
                setattr(P, k, float(v))
        rc = self._L.zo_camera_update(self._h, C.byref(P))
        self.params = p
        if rc != 0:
            raise OracleError(rc)
        return self

    def reset_rng(self):
        self._L.zo_camera_reset_rng(self._h)

    @property
    def rng_state(self):
        r = self._L.zo_camera_rng(self._h).contents
        return (r.x, r.y, r.z, r.w)

    def create_rays(self, samples, rng_states=None, want_first_retry_states=False, threads=0, out=None):
        """samples: (n,4) float32 (sx, sy, lensx, lensy).  Returns dict of planes + flags.
        out: (planes (7,n) float32, flags (n,) uint8) to write into (timing runs reuse page-touched buffers)."""
        s = np.ascontiguousarray(samples, dtype=np.float32)
        n = s.shape[0]
        if out is not None:
            planes, flags = out
            assert planes.shape == (7, n) and planes.dtype == np.float32 and planes.flags.c_contiguous
            assert flags.shape == (n,) and flags.dtype == np.uint8
        else:
            planes = np.zeros((7, n), dtype=np.float32)
            flags = np.zeros(n, dtype=np.uint8)
        rs = None
        if rng_states is not None:
            rs = np.ascontiguousarray(rng_states, dtype=np.uint32)
            assert rs.shape == (n, 4)
        frs = np.zeros((n, 4), dtype=np.uint32) if want_first_retry_states else None
        if threads and threads > 1:
            assert rs is not None, "multi-threaded oracle needs per-ray rng states"
            self._L.zo_create_rays_mt(self._h, n, s.ctypes.data, planes.ctypes.data, flags.ctypes.data,
                                      rs.ctypes.data, int(threads))
        else:
            self._L.zo_create_rays(self._h, n, s.ctypes.data, planes.ctypes.data, flags.ctypes.data,
                                   rs.ctypes.data if rs is not None else None,
                                   frs.ctypes.data if frs is not None else None)
        out = dict(origin=planes[0:3], dir=planes[3:6], weight=planes[6], flags=flags, planes=planes,
                   tries=(flags >> 1).astype(np.int32))
        if frs is not None:
            out["first_retry_states"] = frs
        return out

    # ---- tables -----------------------------------------------------------
    def lens_table(self):
        n = self._L.zo_lens_count(self._h)
        le = self._L.zo_lenses(self._h)
        arr = np.array([[le[i].curvature, le[i].thickness, le[i].ior, le[i].aperture, le[i].center]
                        for i in range(n)], dtype=np.float32).reshape(n, 5)
        return dict(lensCount=n, apertureElement=self._L.zo_aperture_element(self._h), elements=arr,
                    userApertureRadius=np.float32(self._L.zo_user_aperture_radius(self._h)),
                    originShift=np.float32(self._L.zo_origin_shift(self._h)),
                    apertureDistance=np.float32(self._L.zo_aperture_distance(self._h)),
                    focalLengthRatio=np.float32(self._L.zo_focal_length_ratio(self._h)),
                    tracedFocalLength=(np.float32(self._L.zo_traced_focal_length(self._h, 0)),
                                       np.float32(self._L.zo_traced_focal_length(self._h, 1))))

    def lut(self):
        n = self._L.zo_lut_size(self._h)
        keys = np.ctypeslib.as_array(self._L.zo_lut_keys(self._h), shape=(32,))[:n].copy()
        boxes = np.ctypeslib.as_array(self._L.zo_lut_boxes(self._h), shape=(32, 4))[:n].copy()  # max.x max.y min.x min.y
        return keys, boxes

    def thinlens(self):
        return dict(fov=np.float32(self._L.zo_fov(self._h)), tan_fov=np.float32(self._L.zo_tan_fov(self._h)),
                    apertureRadius=np.float32(self._L.zo_aperture_radius(self._h)))

    def counters(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._L.zo_counters(self._h, C.byref(a), C.byref(b), C.byref(c))
        return dict(succesRays=a.value, vignettedRays=b.value, totalInternalReflection=c.value)

    def surface_visits(self):
        """Interfaces entered by traceThroughLensElements since the last lens rebuild (work measure, not a reference counter)."""
        return int(self._L.zo_surface_visits(self._h))

    def bokeh_tables(self):
        x, y = C.c_int(), C.c_int()
        if not self._L.zo_bokeh_dims(self._h, C.byref(x), C.byref(y)):
            return None
        x, y = x.value, y.value
        aa = np.ctypeslib.as_array
        return dict(x=x, y=y,
                    cdfRow=aa(self._L.zo_bokeh_cdf_row(self._h), shape=(y,)).copy(),
                    rowIndices=aa(self._L.zo_bokeh_row_indices(self._h), shape=(y,)).copy(),
                    cdfColumn=aa(self._L.zo_bokeh_cdf_column(self._h), shape=(x * y,)).copy(),
                    columnIndices=aa(self._L.zo_bokeh_column_indices(self._h), shape=(x * y,)).copy())

    def bokeh_sample(self, u1, u2):
        dx, dy = C.c_float(), C.c_float()
        self._L.zo_bokeh_sample(self._h, u1, u2, C.byref(dx), C.byref(dy))
        return np.float32(dx.value), np.float32(dy.value)

    def trace_record(self, origin, direction):
        o, d = V3(*origin), V3(*direction)
        hits = (V3 * 64)()
        nh = C.c_int(0)
        ok = self._L.zo_trace_record(self._h, C.byref(o), C.byref(d), hits, C.byref(nh))
        return bool(ok), np.array([[h.x, h.y, h.z] for h in hits[:nh.value]], dtype=np.float32).reshape(-1, 3), \
            np.array([o.x, o.y, o.z], np.float32), np.array([d.x, d.y, d.z], np.float32)


def concentric_disk_sample(u, v):
    r = V2()
    lib().zo_concentric_disk_sample(u, v, C.byref(r))
    return np.float32(r.x), np.float32(r.y)


def fast_sin(x):
    return np.float32(lib().zo_fast_sin(x))


def fast_cos(x):
    return np.float32(lib().zo_fast_cos(x))


def xor128_stream(n, state=None):
    """First n outputs of the reference xorshift128 from its fixed seed (zoic.cpp:647-652)."""
    r = Rng()
    if state is None:
        lib().zo_rng_seed(C.byref(r))
    else:
        r.x, r.y, r.z, r.w = state
    return np.array([lib().zo_xor128(C.byref(r)) for _ in range(n)], dtype=np.uint32)
