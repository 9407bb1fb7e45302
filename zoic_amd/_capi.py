"""ctypes declarations for libzoic_amd.so -- exactly the symbols include/zoic_amd.h declares.

Loading never falls back to anything else: a missing library, or a missing symbol, raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# ZOIC_AMD_LIB points at another build of the same library (A/B experiments, tools/); there is still no fallback
LIB_PATH = os.environ.get("ZOIC_AMD_LIB") or os.path.join(HERE, "libzoic_amd.so")

ABI_VERSION = 5
MAX_LENS_SURFACES = 32
LUT_ENTRIES = 32

THINLENS, RAYTRACED, LENS_NONE = 0, 1, 2
PRECISION_STRICT, PRECISION_FAST, PRECISION_FAST_UNCHECKED = 0, 1, 2
FRAME_RECORDS, FRAME_PAYLOAD, FRAME_PAYLOAD_SPARSE, FRAME_PAYLOAD_AUTO = 0, 1, 2, 3
TILE_MAX_SAMPLES = 65536

STATUS_NAMES = ["ZOIC_OK", "ZOIC_ERR_INVALID_ARGUMENT", "ZOIC_ERR_LENS_PATH", "ZOIC_ERR_LENS_COLUMNS",
                "ZOIC_ERR_LENS_PARSE", "ZOIC_ERR_MULTI_APERTURE", "ZOIC_ERR_NO_APERTURE", "ZOIC_ERR_TOO_MANY_LENSES",
                "ZOIC_ERR_BOKEH_IMAGE", "ZOIC_ERR_NOT_UPDATED", "ZOIC_ERR_HIP", "ZOIC_ERR_NO_DEVICE"]


class Params(C.Structure):
    _fields_ = [("sensorWidth", C.c_float), ("sensorHeight", C.c_float), ("focalLength", C.c_float),
                ("fStop", C.c_float), ("focalDistance", C.c_float), ("useImage", C.c_int32),
                ("bokehPath", C.c_char_p), ("lensModel", C.c_int32), ("lensDataPath", C.c_char_p),
                ("kolbSamplingLUT", C.c_int32), ("useDof", C.c_int32), ("opticalVignettingDistance", C.c_float),
                ("opticalVignettingRadius", C.c_float), ("exposureControl", C.c_float)]


class CameraInput(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("sx", "sy", "dsx", "dsy", "lensx", "lensy", "relative_time")]


class Vec3(C.Structure):
    _fields_ = [(n, C.c_float) for n in "xyz"]


class CameraOutput(C.Structure):
    _fields_ = [("origin", Vec3), ("dir", Vec3), ("dOdx", Vec3), ("dOdy", Vec3), ("dDdx", Vec3), ("dDdy", Vec3),
                ("weight", C.c_float * 3)]


class Ray(C.Structure):   # zoic_ray: one 32-byte record per camera ray
    _fields_ = [(n, C.c_float) for n in ("ox", "oy", "oz", "dx", "dy", "dz", "weight")] + [("flags", C.c_uint32)]


RAY_DTYPE = [("ox", "<f4"), ("oy", "<f4"), ("oz", "<f4"), ("dx", "<f4"), ("dy", "<f4"), ("dz", "<f4"), ("weight", "<f4"),
             ("flags", "<u4")]


class FrameLaneInfo(C.Structure):
    _fields_ = [("device", C.c_int32), ("peer_access_to_root", C.c_int32), ("peer_access_from_root", C.c_int32), ("chunks", C.c_uint32),
                ("rays", C.c_uint64), ("bytes_to_root", C.c_uint64)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("succesRays", "vignettedRays", "totalInternalReflection")]


class LensInfo(C.Structure):
    _fields_ = ([("lensCount", C.c_int32), ("apertureElement", C.c_int32), ("userApertureRadius", C.c_float),
                 ("originShift", C.c_float), ("apertureDistance", C.c_float), ("focalLengthRatio", C.c_float),
                 ("tracedFocalLength", C.c_float * 2), ("fov", C.c_float), ("tan_fov", C.c_float),
                 ("apertureRadius", C.c_float)]
                + [(n, C.c_float * MAX_LENS_SURFACES) for n in ("curvature", "thickness", "ior", "aperture", "center")]
                + [("lutSize", C.c_int32), ("lutKey", C.c_float * LUT_ENTRIES)]
                + [(n, C.c_float * LUT_ENTRIES) for n in ("lutMaxX", "lutMaxY", "lutMinX", "lutMinY")]
                + [("bokehWidth", C.c_int32), ("bokehHeight", C.c_int32), ("fastRunsStrict", C.c_int32), ("precomputeTIR", C.c_uint32)])


# every symbol include/zoic_amd.h declares: name -> (restype, argtypes)
_vp, _u64, _u32 = C.c_void_p, C.c_uint64, C.c_uint32
SYMBOLS = {
    "zoic_abi_version": (C.c_int, []),
    "zoic_status_string": (C.c_char_p, [C.c_int]),
    "zoic_last_error_string": (C.c_char_p, []),
    "zoic_device_count": (C.c_int, []),
    "zoic_device_numa_node": (C.c_int, [C.c_int]),
    "zoic_params_default": (None, [C.POINTER(Params)]),
    "zoic_camera_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "zoic_camera_destroy": (None, [_vp]),
    "zoic_camera_update": (C.c_int, [_vp, C.POINTER(Params)]),
    "zoic_camera_set_bokeh_image": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    "zoic_camera_set_lens_text": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "zoic_camera_set_precision": (C.c_int, [_vp, C.c_int]),
    "zoic_camera_set_frame_aspect": (C.c_int, [_vp, C.c_float]),
    "zoic_camera_set_wait_mode": (C.c_int, [_vp, C.c_int]),
    "zoic_camera_set_seed": (C.c_int, [_vp, _u32]),
    "zoic_create_rays_device": (C.c_int, [_vp, _u64, _vp, _vp, _u64, _vp, _vp]),
    "zoic_create_rays_host": (C.c_int, [_vp, _u64, _vp, _vp, _u64, _vp]),
    "zoic_create_rays_arnold": (C.c_int, [_vp, _u64, C.POINTER(CameraInput), C.POINTER(CameraOutput), _u64]),
    "zoic_camera_create_ray": (C.c_int, [_vp, C.POINTER(CameraInput), C.POINTER(CameraOutput), C.c_uint16]),
    "zoic_tile_create": (C.c_int, [_vp, _u32, C.c_uint16, C.POINTER(_vp)]),
    "zoic_tile_destroy": (None, [_vp]),
    "zoic_tile_inputs": (C.POINTER(CameraInput), [_vp]),
    "zoic_tile_outputs": (C.POINTER(CameraOutput), [_vp]),
    "zoic_tile_capacity": (_u32, [_vp]),
    "zoic_tile_submit": (C.c_int, [_vp, _u32, _u64]),
    "zoic_tile_wait": (C.c_int, [_vp]),
    "zoic_tile_done": (C.c_int, [_vp]),
    "zoic_create_rays_device_resident": (C.c_int, [_vp, C.c_uint32, _vp, _vp, C.c_uint64, C.c_uint16]),
    "zoic_tile_set_rows": (C.c_int, [_vp, C.c_int]),
    "zoic_tile_rays": (C.c_void_p, [_vp]),
    "zoic_tile_set_inputs": (C.c_int, [_vp, C.c_int]),
    "zoic_tile_samples": (C.c_void_p, [_vp]),
    "zoic_camera_create_rays_tile": (C.c_int, [_vp, _u32, C.POINTER(CameraInput), C.POINTER(CameraOutput), _u64, C.c_uint16]),
    "zoic_camera_reverse_ray": (C.c_int, [_vp, C.POINTER(Vec3), C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "zoic_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(_vp)]),
    "zoic_host_free": (None, [_vp]),
    "zoic_host_register": (C.c_int, [_vp, C.c_size_t]),
    "zoic_host_unregister": (C.c_int, [_vp]),
    "zoic_generate_samples_device": (C.c_int, [_vp, _u64, _u64, _u32, _u32, _u32, _u32, _vp, _vp]),
    "zoic_frame_slab": (C.c_int, [_u64, C.c_int, C.c_int, C.POINTER(_u64), C.POINTER(_u64)]),
    "zoic_frame_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_vp)]),
    "zoic_frame_destroy": (None, [_vp]),
    "zoic_frame_device_count": (C.c_int, [_vp]),
    "zoic_frame_camera": (_vp, [_vp, C.c_int]),
    "zoic_frame_set_bokeh_image": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    "zoic_frame_set_lens_text": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "zoic_frame_set_precision": (C.c_int, [_vp, C.c_int]),
    "zoic_frame_set_seed": (C.c_int, [_vp, _u32]),
    "zoic_frame_update": (C.c_int, [_vp, C.POINTER(Params)]),
    "zoic_frame_set_chunk_rays": (C.c_int, [_vp, _u64]),
    "zoic_frame_render_device": (C.c_int, [_vp, _u64, C.POINTER(_vp), _u64, _vp, C.c_int, _vp]),
    "zoic_frame_render_local": (C.c_int, [_vp, _u64, C.POINTER(_vp), _u64, C.POINTER(_vp)]),
    "zoic_frame_render_host": (C.c_int, [_vp, _u64, _vp, _u64, _vp]),
    "zoic_frame_generate_samples": (C.c_int, [_vp, _u64, _u64, _u32, _u32, _u32, _u32]),
    "zoic_frame_synchronize": (C.c_int, [_vp]),
    "zoic_frame_get_lane_info": (C.c_int, [_vp, C.c_int, C.POINTER(FrameLaneInfo)]),
    "zoic_frame_get_counters": (C.c_int, [_vp, C.POINTER(Counters)]),
    "zoic_frame_auto_layout": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "zoic_camera_get_counters": (C.c_int, [_vp, C.POINTER(Counters)]),
    "zoic_camera_reset_counters": (C.c_int, [_vp]),
    "zoic_camera_get_info": (C.c_int, [_vp, C.POINTER(LensInfo)]),
    "zoic_camera_get_bokeh_tables": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
}

_lib = None


class ZoicLibraryError(RuntimeError):
    pass


def load(path=None):
    """dlopen libzoic_amd.so and bind every declared symbol.  No fallback of any kind."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ZoicLibraryError("libzoic_amd.so is not built (%s): run `python -m zoic_amd.build`; "
                               "there is no CPU fallback" % p)
    try:
        # torch (if it is going to be used in this process) ships its own libamdhip64.so.7; importing it first
        # makes both share ONE HIP runtime (same SONAME), so torch device pointers are valid here.
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise ZoicLibraryError("libzoic_amd.so lacks symbol %s declared in include/zoic_amd.h" % name)
        fn.restype = res
        fn.argtypes = args
    if lib.zoic_abi_version() != ABI_VERSION:
        raise ZoicLibraryError("ABI version mismatch")
    if path is None:
        _lib = lib
    return lib
