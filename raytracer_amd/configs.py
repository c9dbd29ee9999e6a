"""The five workloads BASELINE.json names, as concrete synthetic inputs.

Camera: reference defaults (camera.rs:59-72). Scenes come from the host mirror's seeded builders.
C3's "glowing checkers" shader is not in the reference at HEAD; RT_SHADE_CHECKER is this build's
documented reconstruction.
"""
from . import _abi, host

CONFIGS = {
    # name: (width, height, spp, frames, scene, flags, description)
    "C1": (640, 360, 1, 64, ("default", 1), _abi.RT_ACCUM_RGBA8,
           "default 4-sphere scene, 640x360, 1 spp x 64 frames (CPU-runnable case)"),
    "C2": (1920, 1080, 16, 256, ("default", 1), _abi.RT_ACCUM_RGBA8,
           "default scene, 1920x1080, 16 spp x 256 frames, one GPU"),
    "C3": (3840, 2160, 64, 8, ("bunch", 20, 3, 3), _abi.RT_ACCUM_F32 | _abi.RT_SHADE_CHECKER,
           "glowing-checkers reconstruction, random_bunch(20), 3840x2160, 64 spp"),
    "C4": (1920, 1080, 4, 16, ("bunch", 1024, 16, 4), _abi.RT_ACCUM_RGBA8,
           "random_bunch(1024), 16 materials, 1920x1080, 4 spp x 16 frames"),
    "C5": (7680, 4320, 1, 1024, ("default", 1), _abi.RT_ACCUM_F32 | _abi.RT_COMBINE_SUM,
           "default scene, 7680x4320, 1 spp x 1024 frames split by frame sets, RCCL reduce"),
}


def scene(spec):
    if spec[0] == "default":
        return host.scene_default(spec[1])
    if spec[0] == "bunch":
        return host.scene_random_bunch(spec[1], spec[2], spec[3])
    raise KeyError(spec)


def get(name):
    w, h, spp, frames, sc, flags, desc = CONFIGS[name]
    spheres, materials = scene(sc)
    return {"name": name, "width": w, "height": h, "spp": spp, "frames": frames, "spheres": spheres,
            "materials": materials, "flags": flags, "description": desc}
