"""ctypes binding of librt_host.so — the headless C++ mirror of the reference's Rust host
(Camera / World / Time / App; include/rt_host.h). Plumbing only: the arithmetic lives in
csrc/host/rt_host.cpp.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from .raytracer import CAMERA_DTYPE, MATERIAL_DTYPE, SPHERE_DTYPE

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librt_host.so")
_VP = C.c_void_p

SIGNATURES = {
    "rth_camera_raw": (C.c_int, [C.c_uint32, C.c_uint32, _VP, _VP, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, _VP]),
    "rth_camera_create": (_VP, [C.c_uint32, C.c_uint32]),
    "rth_camera_destroy": (None, [_VP]),
    "rth_camera_key": (C.c_int, [_VP, C.c_char_p]),
    "rth_camera_zoom": (C.c_int, [_VP, C.c_float]),
    "rth_camera_look_at_pixel": (C.c_int, [_VP, C.c_float, C.c_float]),
    "rth_camera_increase_frame": (C.c_int, [_VP]),
    "rth_camera_render_finished": (C.c_int, [_VP]),
    "rth_camera_get": (C.c_int, [_VP, _VP, C.POINTER(C.c_float)]),
    "rth_scene_default": (C.c_int, [C.c_uint64, _VP, _VP]),
    "rth_scene_random_bunch": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint64, _VP, _VP]),
    "rth_time_table": (C.c_int, [C.c_uint32, C.c_uint32, _VP]),
    "rth_app_create": (C.c_int, [C.POINTER(_VP), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, _VP, C.c_uint32, _VP,
                                 C.c_uint32, C.c_uint32, C.c_uint32]),
    "rth_app_destroy": (None, [_VP]),
    "rth_app_redraw": (C.c_int, [_VP, C.c_uint32]),
    "rth_app_run": (C.c_int, [_VP, C.c_uint32]),
    "rth_app_ctx": (_VP, [_VP]),
    "rth_app_current_frame": (C.c_uint32, [_VP]),
    "rth_app_time_bits": (C.c_uint32, [_VP]),
    "rth_app_camera_key": (C.c_int, [_VP, C.c_char_p]),
    "rth_app_camera_zoom": (C.c_int, [_VP, C.c_float]),
    "rth_app_camera_look_at_pixel": (C.c_int, [_VP, C.c_float, C.c_float]),
    "rth_app_camera_get": (C.c_int, [_VP, _VP, C.POINTER(C.c_float)]),
    "rth_app_save_checkpoint": (C.c_int, [_VP, C.c_char_p]),
    "rth_app_load_checkpoint": (C.c_int, [_VP, C.c_char_p]),
    "rth_checkpoint_info": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rth_write_png": (C.c_int, [C.c_char_p, _VP, C.c_uint32, C.c_uint32, C.c_int]),
    "rth_last_error": (C.c_char_p, []),
}

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    _abi.load()  # librt_host.so links against librt_hip.so
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc`")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise _abi.RtError(rc, load().rth_last_error().decode("utf-8", "replace"))


def _p(a):
    return None if a is None else a.ctypes.data_as(_VP)


def camera_raw(width, height, spp=1, frames=8, current_frame=0, pos=None, look=None, fov=0.0):
    """Camera::to_raw for a headless camera (reference camera.rs:77-131)."""
    out = np.zeros(1, CAMERA_DTYPE)
    p = None if pos is None else np.asarray(pos, np.float32)
    l = None if look is None else np.asarray(look, np.float32)
    _check(load().rth_camera_raw(width, height, _p(p), _p(l), fov, spp, frames, current_frame, _p(out)))
    return out[0]


def scene_default(seed=1):
    """World::new (reference world.rs:34-40) with seeded materials."""
    s, m = np.zeros(4, SPHERE_DTYPE), np.zeros(3, MATERIAL_DTYPE)
    _check(load().rth_scene_default(seed, _p(s), _p(m)))
    return s, m


def scene_random_bunch(amount, n_materials, seed):
    """Sphere::random_bunch (reference objects.rs:27-40) with in-range material indices."""
    s, m = np.zeros(amount, SPHERE_DTYPE), np.zeros(n_materials, MATERIAL_DTYPE)
    _check(load().rth_scene_random_bunch(amount, n_materials, seed, _p(s), _p(m)))
    return s, m


def time_table(n, start_bits=0):
    """time bits of redraws 0..n-1: repeated f32 += 0.01 (reference time.rs:72-77, app.rs:94)."""
    out = np.zeros(n, np.uint32)
    _check(load().rth_time_table(start_bits, n, _p(out)))
    return out


def checkpoint_info(path):
    """Header of a checkpoint file (no GPU needed)."""
    v = (C.c_uint32 * 9)()
    h, n = C.c_uint64(), C.c_uint64()
    _check(load().rth_checkpoint_info(os.fsencode(path), v, C.byref(h), C.byref(n)))
    keys = ("width", "height", "flags", "current_frame", "frames_to_render", "samples_per_pixel", "time_bits", "n_spheres", "n_materials")
    d = dict(zip(keys, (int(x) for x in v)))
    d["scene_hash"], d["image_bytes"] = int(h.value), int(n.value)
    return d


def write_png(path, rgba8, present=False):
    a = np.ascontiguousarray(rgba8, np.uint8)
    _check(load().rth_write_png(os.fsencode(path), _p(a), a.shape[1], a.shape[0], int(present)))


class Camera:
    """Stateful headless Camera (reference camera.rs:23-288): keys, zoom, look-at-pixel, frame counter."""

    def __init__(self, width, height):
        self._lib = load()
        self._h = _VP(self._lib.rth_camera_create(width, height))

    def key(self, name):
        _check(self._lib.rth_camera_key(self._h, name.encode()))

    def zoom(self, val):
        _check(self._lib.rth_camera_zoom(self._h, val))

    def look_at_pixel(self, off_x, off_y):
        _check(self._lib.rth_camera_look_at_pixel(self._h, off_x, off_y))

    def increase_frame(self):
        _check(self._lib.rth_camera_increase_frame(self._h))

    def render_finished(self):
        return bool(self._lib.rth_camera_render_finished(self._h))

    def raw(self):
        out = np.zeros(1, CAMERA_DTYPE)
        fov = C.c_float()
        _check(self._lib.rth_camera_get(self._h, _p(out), C.byref(fov)))
        return out[0], fov.value

    def __del__(self):
        try:
            self._lib.rth_camera_destroy(self._h)
        except Exception:
            pass


class App:
    """App::new / update / render (reference app.rs:26-48,93-116), compute pass only."""

    def __init__(self, width, height, flags, spheres, materials, spp, frames, device=0):
        self._lib = load()
        self._h = _VP()
        self.width, self.height = width, height
        s = np.ascontiguousarray(spheres, SPHERE_DTYPE)
        m = np.ascontiguousarray(materials, MATERIAL_DTYPE)
        _check(self._lib.rth_app_create(C.byref(self._h), device, width, height, flags, _p(s), s.shape[0], _p(m),
                                        m.shape[0], spp, frames))

    def redraw(self, n=1):
        _check(self._lib.rth_app_redraw(self._h, n))

    def run(self, n):
        _check(self._lib.rth_app_run(self._h, n))

    @property
    def current_frame(self):
        return self._lib.rth_app_current_frame(self._h)

    @property
    def time_bits(self):
        return self._lib.rth_app_time_bits(self._h)

    # the interactive controls on the App's own camera (each ends with scene_was_updated: frame 0, camera uploaded)
    def key(self, name):
        _check(self._lib.rth_app_camera_key(self._h, name.encode()))

    def zoom(self, val):
        _check(self._lib.rth_app_camera_zoom(self._h, val))

    def look_at_pixel(self, off_x, off_y):
        _check(self._lib.rth_app_camera_look_at_pixel(self._h, off_x, off_y))

    def camera_raw(self):
        out = np.zeros(1, CAMERA_DTYPE)
        _check(self._lib.rth_app_camera_get(self._h, _p(out), None))
        return out[0]

    def read_rng_state(self):
        out = np.empty((self.height, self.width), np.uint32)
        _abi.check(_abi.load().rt_read_rng_state(self._lib.rth_app_ctx(self._h), _p(out), out.nbytes))
        return out

    def enable_rng_capture(self, on=True):
        _abi.check(_abi.load().rt_enable_rng_capture(self._lib.rth_app_ctx(self._h), int(on)))

    def read_rgba8(self):
        out = np.empty((self.height, self.width, 4), np.uint8)
        _abi.check(_abi.load().rt_read_rgba8(self._lib.rth_app_ctx(self._h), _p(out), out.nbytes))
        return out

    def save_checkpoint(self, path):
        _check(self._lib.rth_app_save_checkpoint(self._h, os.fsencode(path)))

    def load_checkpoint(self, path):
        _check(self._lib.rth_app_load_checkpoint(self._h, os.fsencode(path)))

    def close(self):
        if self._h:
            self._lib.rth_app_destroy(self._h)
            self._h = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
