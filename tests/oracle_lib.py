"""ctypes binding of oracle/librt_oracle.so — the CPU checker (test infrastructure only).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by
raytracer_amd/. Builds the library with oracle/Makefile when it is missing (gcc only).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "librt_oracle.so")

SPHERE_DTYPE = np.dtype(
    [("center", "<f4", 3), ("padding", "<u4"), ("radius", "<f4"), ("material_index", "<u4"), ("pad1", "<u4", 2)]
)
MATERIAL_DTYPE = np.dtype(
    [("color", "<f4", 4), ("metallic", "<f4"), ("roughness", "<f4"), ("specular", "<f4"), ("padding", "<f4")]
)
CAMERA_DTYPE = np.dtype(
    [
        ("world_to_pixel", "<f4", 16),
        ("pixel_to_world", "<f4", 16),
        ("pos", "<f4", 4),
        ("focal_length", "<f4"),
        ("samples_per_pixel", "<u4"),
        ("frames_to_render", "<u4"),
        ("current_frame", "<u4"),
    ]
)
COUNTER_NAMES = ("T", "D", "H", "R", "M", "S", "Sm", "J", "Sd", "N", "P", "rand_calls")

ACCUM_F32 = 0x1
DISPATCH_FULL = 0x2
SHADE_CHECKER = 0x4
COMBINE_SUM = 0x10

_lib = None


def build():
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    VP = C.c_void_p
    lib.rto_compute.restype = C.c_int
    lib.rto_compute.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32,
                                VP, VP, VP, VP, C.c_uint32, C.c_uint32, C.c_int]
    lib.rto_render.restype = C.c_int
    lib.rto_render.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, VP, VP, C.c_uint32,
                               VP, VP, VP, VP, C.c_uint32, C.c_uint32, C.c_int]
    lib.rto_time_advance.restype = C.c_uint32
    lib.rto_time_advance.argtypes = [C.c_uint32, C.c_uint32]
    lib.rto_rng_seed.restype = C.c_uint32
    lib.rto_rng_seed.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    lib.rto_rng_step.restype = C.c_uint32
    lib.rto_rng_step.argtypes = [C.c_uint32, C.POINTER(C.c_float)]
    lib.rto_sizeof.restype = C.c_uint32
    lib.rto_sizeof.argtypes = [C.c_int]
    lib.rto_max_threads.restype = C.c_int
    _lib = lib
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleImage:
    """In/out state of the oracle's accumulation image (one per render)."""

    def __init__(self, width, height, flags):
        self.width, self.height, self.flags = width, height, flags
        self.rgba8 = np.zeros((height, width, 4), np.uint8)
        self.accum = np.zeros((height, width, 4), np.float32) if flags & ACCUM_F32 else None
        self.rng = np.zeros((height, width), np.uint32)
        self.counters = np.zeros(12, np.uint64)

    def counters_dict(self):
        return {n: int(v) for n, v in zip(COUNTER_NAMES, self.counters)}

    def view_rgba8(self):
        """RGBA8 view: the image itself, or the float image under the rgba8unorm store rule."""
        if self.accum is None:
            return self.rgba8
        c = np.clip(np.nan_to_num(self.accum, nan=0.0), 0.0, 1.0).astype(np.float32)
        return np.rint(c * np.float32(255.0)).astype(np.uint8)


def compute(img, spheres, materials, camera, time_bits, rank=0, world=1, threads=0):
    lib = load()
    s = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    m = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
    c = np.ascontiguousarray(camera, dtype=CAMERA_DTYPE).reshape(1)
    rc = lib.rto_compute(img.width, img.height, img.flags, _p(s), s.shape[0], _p(m), m.shape[0], _p(c),
                         int(time_bits), _p(img.rgba8), _p(img.accum), _p(img.rng), _p(img.counters), rank, world, threads)
    if rc:
        raise ValueError("rto_compute rejected its arguments")


def render(img, spheres, materials, camera, time_bits, n_frames, rank=0, world=1, threads=0):
    """App loop for n_frames redraws. Returns (camera_after, time_bits_after)."""
    lib = load()
    s = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    m = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
    c = np.array(camera, dtype=CAMERA_DTYPE).reshape(1).copy()
    t = np.array([time_bits], np.uint32)
    rc = lib.rto_render(img.width, img.height, img.flags, _p(s), s.shape[0], _p(m), m.shape[0], _p(c), _p(t), n_frames,
                        _p(img.rgba8), _p(img.accum), _p(img.rng), _p(img.counters), rank, world, threads)
    if rc:
        raise ValueError("rto_render rejected its arguments")
    return c[0], int(t[0])


def time_bits_table(n, start_bits=0):
    """time bits for frames 0..n-1 of a fresh process: repeated f32 += 0.01 (time.rs:72-77)."""
    lib = load()
    out = np.empty(n, np.uint32)
    b = start_bits
    for i in range(n):
        b = lib.rto_time_advance(b, 1)
        out[i] = b
    return out


def algorithmic_flops(c):
    """FLOPs = 17T + 4D + 12H + 6R + 15M + 8S + 12Sm + 11J + 7Sd + 8N + 110P."""
    return (17 * c["T"] + 4 * c["D"] + 12 * c["H"] + 6 * c["R"] + 15 * c["M"] + 8 * c["S"] + 12 * c["Sm"]
            + 11 * c["J"] + 7 * c["Sd"] + 8 * c["N"] + 110 * c["P"])
