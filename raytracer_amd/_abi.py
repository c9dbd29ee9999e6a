"""ctypes binding of librt_hip.so — exactly the entry points declared in include/rt_hip.h.

This module is plumbing: it loads the in-tree shared library and declares argument types. It
contains no compute and no fallback; if the library is missing the import fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_hip.so")

# flags — numeric values of include/rt_types.h
RT_ACCUM_RGBA8 = 0x0
RT_ACCUM_F32 = 0x1
RT_DISPATCH_FLOOR16 = 0x0
RT_DISPATCH_FULL = 0x2
RT_SHADE_HEAD = 0x0
RT_SHADE_CHECKER = 0x4
RT_MATH_STRICT = 0x0
RT_MATH_FAST = 0x8
RT_COMBINE_AVERAGE = 0x0
RT_COMBINE_SUM = 0x10

RT_OK = 0
RT_ERR_INVALID = -1
RT_ERR_HIP = -2
RT_ERR_STATE = -3
RT_ERR_NO_DEVICE = -4


class RtCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("T", "D", "H", "R", "M", "S", "Sm", "J", "Sd", "N", "P", "rand_calls", "slots_trace", "slots_reject")]

    EVENTS = ("T", "D", "H", "R", "M", "S", "Sm", "J", "Sd", "N", "P", "rand_calls")

    def as_dict(self):
        """The twelve event counters the oracle also keeps."""
        return {n: int(getattr(self, n)) for n in self.EVENTS}

    def occupancy(self):
        """Lane-occupancy diagnostics of the kernel."""
        return {"slots_trace": int(self.slots_trace), "slots_reject": int(self.slots_reject)}


# name -> (restype, argtypes); every symbol include/rt_hip.h declares
_VP = C.c_void_p
SIGNATURES = {
    "rt_create": (C.c_int, [C.POINTER(_VP), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rt_destroy": (None, [_VP]),
    "rt_set_scene": (C.c_int, [_VP, _VP, C.c_uint32, _VP, C.c_uint32]),
    "rt_set_camera": (C.c_int, [_VP, _VP]),
    "rt_set_time": (C.c_int, [_VP, C.c_uint32]),
    "rt_compute": (C.c_int, [_VP]),
    "rt_render_frames": (C.c_int, [_VP, _VP, C.c_uint32]),
    "rt_render_frames_static": (C.c_int, [_VP, _VP, C.c_uint32]),
    "rt_flush": (C.c_int, [_VP]),
    "rt_sync": (C.c_int, [_VP]),
    "rt_read_rgba8": (C.c_int, [_VP, _VP, C.c_size_t]),
    "rt_read_accum_f32": (C.c_int, [_VP, _VP, C.c_size_t]),
    "rt_write_image": (C.c_int, [_VP, _VP, C.c_size_t]),
    "rt_enable_rng_capture": (C.c_int, [_VP, C.c_int]),
    "rt_read_rng_state": (C.c_int, [_VP, _VP, C.c_size_t]),
    "rt_enable_counters": (C.c_int, [_VP, C.c_int]),
    "rt_get_counters": (C.c_int, [_VP, C.POINTER(RtCounters)]),
    "rt_reset_counters": (C.c_int, [_VP]),
    "rt_set_partition": (C.c_int, [_VP, C.c_uint32, C.c_uint32]),
    "rt_sky_tile_stats": (C.c_int, [_VP, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rt_accum_device_ptr": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(C.c_size_t)]),
    "rt_bind_accum": (C.c_int, [_VP, _VP, C.c_size_t]),
    "rt_set_stream": (C.c_int, [_VP, _VP]),
    "rt_clear": (C.c_int, [_VP]),
    "rt_enable_timing": (C.c_int, [_VP, C.c_int]),
    "rt_timing_reset": (C.c_int, [_VP]),
    "rt_timing_get": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "rt_device_info": (C.c_int, [_VP, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rt_last_error": (C.c_char_p, []),
    "rt_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load librt_hip.so (once). Raises if it has not been built — there is no other path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
            "(or __graft_entry__.build()). raytracer_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code


def check(code):
    if code != RT_OK:
        raise RtError(code, load().rt_last_error().decode("utf-8", "replace"))
