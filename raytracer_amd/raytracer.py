"""Thin object wrapper over the C ABI, named after the reference seam it stands in for:
`Raytracer::new` / `Raytracer::compute` (reference src/raytracing.rs:16,178).

Plumbing only — every method is one call into librt_hip.so. Arrays cross the boundary as numpy
buffers in the reference's Pod layouts (Sphere 32 B, Material 32 B, CameraRaw 160 B).
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import check

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
assert SPHERE_DTYPE.itemsize == 32 and MATERIAL_DTYPE.itemsize == 32 and CAMERA_DTYPE.itemsize == 160


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Raytracer:
    """One accumulation image on one GPU. `flags` are the RT_* bits of include/rt_types.h."""

    def __init__(self, width, height, flags=0, device=0):
        self._lib = _abi.load()
        self._h = C.c_void_p()
        self.width, self.height, self.flags, self.device = int(width), int(height), int(flags), int(device)
        check(self._lib.rt_create(C.byref(self._h), device, width, height, flags))

    # Raytracer::new is the constructor; keep the reference's spelling available too
    @classmethod
    def new(cls, width, height, flags=0, device=0):
        return cls(width, height, flags, device)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- inputs -------------------------------------------------------------------------------
    def set_scene(self, spheres, materials):
        s = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        m = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        check(self._lib.rt_set_scene(self._h, _ptr(s), s.shape[0], _ptr(m), m.shape[0]))

    def set_camera(self, camera_raw):
        c = np.ascontiguousarray(camera_raw, dtype=CAMERA_DTYPE).reshape(1)
        check(self._lib.rt_set_camera(self._h, _ptr(c)))

    def set_time(self, time_bits):
        check(self._lib.rt_set_time(self._h, int(time_bits) & 0xFFFFFFFF))

    def set_partition(self, rank, world):
        check(self._lib.rt_set_partition(self._h, rank, world))

    def sky_tile_stats(self):
        """(tiles the last pass finished without tracing, tiles of this context's partition)"""
        a, b = C.c_uint32(0), C.c_uint32(0)
        check(self._lib.rt_sky_tile_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def set_stream(self, hip_stream_handle):
        check(self._lib.rt_set_stream(self._h, C.c_void_p(hip_stream_handle)))

    # -- work ---------------------------------------------------------------------------------
    def compute(self):
        check(self._lib.rt_compute(self._h))

    def render_frames(self, time_bits):
        t = np.ascontiguousarray(time_bits, dtype=np.uint32)
        check(self._lib.rt_render_frames(self._h, _ptr(t), t.shape[0]))

    def render_frames_static(self, time_bits):
        """rt_render_frames without any allocation (static device scratch; include/rt_hip.h has the limits)."""
        t = np.ascontiguousarray(time_bits, dtype=np.uint32)
        check(self._lib.rt_render_frames_static(self._h, _ptr(t), t.shape[0]))

    def flush(self):
        """Launch recorded passes without waiting (queue.submit)."""
        check(self._lib.rt_flush(self._h))

    def sync(self):
        check(self._lib.rt_sync(self._h))

    def clear(self):
        check(self._lib.rt_clear(self._h))

    # -- outputs ------------------------------------------------------------------------------
    def read_rgba8(self):
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        check(self._lib.rt_read_rgba8(self._h, _ptr(out), out.nbytes))
        return out

    def read_accum_f32(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        check(self._lib.rt_read_accum_f32(self._h, _ptr(out), out.nbytes))
        return out

    def write_image(self, image):
        """Upload a checkpointed image (uint8 HxWx4 for RGBA8 contexts, float32 HxWx4 for F32)."""
        a = np.ascontiguousarray(image)
        check(self._lib.rt_write_image(self._h, _ptr(a), a.nbytes))

    def enable_rng_capture(self, on=True):
        check(self._lib.rt_enable_rng_capture(self._h, int(on)))

    def read_rng_state(self):
        out = np.empty((self.height, self.width), dtype=np.uint32)
        check(self._lib.rt_read_rng_state(self._h, _ptr(out), out.nbytes))
        return out

    def enable_counters(self, on=True):
        """True / 1: the reference algorithm's events; 2: the events the production launch executes (include/rt_hip.h)."""
        check(self._lib.rt_enable_counters(self._h, int(on)))

    def reset_counters(self):
        check(self._lib.rt_reset_counters(self._h))

    def counters(self):
        c = _abi.RtCounters()
        check(self._lib.rt_get_counters(self._h, C.byref(c)))
        return c.as_dict()

    def occupancy(self):
        c = _abi.RtCounters()
        check(self._lib.rt_get_counters(self._h, C.byref(c)))
        return c.occupancy()

    def accum_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        check(self._lib.rt_accum_device_ptr(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def bind_accum(self, device_ptr, nbytes):
        check(self._lib.rt_bind_accum(self._h, C.c_void_p(device_ptr), nbytes))

    def enable_timing(self, on=True):
        check(self._lib.rt_enable_timing(self._h, int(on)))

    def timing_reset(self):
        check(self._lib.rt_timing_reset(self._h))

    def timing(self):
        ms, n = C.c_double(), C.c_uint32()
        check(self._lib.rt_timing_get(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def device_info(self):
        name = C.create_string_buffer(256)
        cu, khz = C.c_int(), C.c_int()
        check(self._lib.rt_device_info(self._h, name, 256, C.byref(cu), C.byref(khz)))
        return {"name": name.value.decode(), "compute_units": cu.value, "clock_khz": khz.value}
