"""zoic_amd -- MI355X-native camera-ray generator for zoic's per-sample lens hot path.

The product is libzoic_amd.so (HIP kernels + C-ABI, include/zoic_amd.h).  This package is the thin host-side
mirror of the reference's Arnold node interface over that C-ABI; it contains no ray arithmetic and no CPU
fallback -- if the library is missing, importing the camera raises.
"""
from ._capi import PRECISION_FAST, PRECISION_FAST_UNCHECKED, PRECISION_STRICT, RAYTRACED, THINLENS, ZoicLibraryError  # noqa: F401
from ._capi import FRAME_PAYLOAD, FRAME_PAYLOAD_AUTO, FRAME_PAYLOAD_SPARSE, FRAME_RECORDS  # noqa: F401
from .camera import DEFAULTS, PinnedArray, ZoicCamera, ZoicError, ZoicTile, lens_path  # noqa: F401
from .frame import ZoicFrame, frame_slab  # noqa: F401
from .placement import pick_frame_buffers  # noqa: F401

__all__ = ["ZoicCamera", "ZoicFrame", "frame_slab", "pick_frame_buffers", "FRAME_RECORDS", "FRAME_PAYLOAD", "FRAME_PAYLOAD_SPARSE", "FRAME_PAYLOAD_AUTO", "ZoicTile", "PinnedArray", "ZoicError", "ZoicLibraryError", "DEFAULTS", "lens_path", "RAYTRACED", "THINLENS",
           "PRECISION_STRICT", "PRECISION_FAST", "PRECISION_FAST_UNCHECKED"]
