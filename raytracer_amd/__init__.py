"""raytracer_amd — MI355X (gfx950) path-tracing compute pass behind a C ABI.

The product is `librt_hip.so` (raytracer_amd/csrc, include/rt_hip.h). This package is the Python
plumbing over it: ctypes signatures (`_abi`), an object wrapper named after the reference seam
(`Raytracer`), the headless host mirror (`host`) and the multi-GPU combine (`dist`).
There is no CPU implementation here; without the built library the imports below raise.
"""
from . import _abi
from ._abi import (  # noqa: F401
    RT_ACCUM_F32,
    RT_ACCUM_RGBA8,
    RT_COMBINE_AVERAGE,
    RT_COMBINE_SUM,
    RT_DISPATCH_FLOOR16,
    RT_DISPATCH_FULL,
    RT_MATH_FAST,
    RT_MATH_STRICT,
    RT_SHADE_CHECKER,
    RT_SHADE_HEAD,
    RtError,
)
from .raytracer import CAMERA_DTYPE, MATERIAL_DTYPE, SPHERE_DTYPE, Raytracer  # noqa: F401

_abi.load()  # fail loudly at import if the HIP library has not been built
