#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small input/output vectors of the compute pass.

The reference ships no fixtures and cannot run here, so these vectors are minted by
this repo's CPU oracle (oracle/rt_oracle.c) from inputs built by the host mirror; they pin the
oracle against drift and let the HIP path be checked on the GPU box without the oracle.
Each file holds the exact input bytes (spheres, materials, CameraRaw, time bits) and the outputs
(RGBA8 image, final RNG state per pixel, float image when applicable, event counters).

    python tests/golden/make_golden.py        # rewrites the fixtures and the C1 checksum
    python tests/golden/make_golden.py --full # also the full-size C2 / C3 / C4 / C5(16 frames) checksums (minutes)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib as O  # noqa: E402
from raytracer_amd import host  # noqa: E402

CASES = {
    # name: (W, H, spp, frames, flags, scene)
    "head_rgba8_64x48": (64, 48, 2, 3, 0, ("default", 1)),
    "head_f32_70x50_full": (70, 50, 2, 2, O.ACCUM_F32 | O.DISPATCH_FULL, ("default", 1)),
    "checker_bunch20_96x64": (96, 64, 4, 2, O.ACCUM_F32 | O.SHADE_CHECKER, ("bunch", 20, 3, 3)),
    "bunch64_48x32": (48, 32, 2, 2, 0, ("bunch", 64, 16, 4)),
}
# whole-config checksums (too big to store): C1 complete
CHECKSUMS = {"C1_640x360_1spp_64frames": (640, 360, 1, 64, 0, ("default", 1))}
# The other BASELINE configs at full size: minutes of CPU each, so only with `--full` (existing entries are kept
# otherwise). C5 is the first 16 of its 1024 frames (8K, frame sum).
FULL_CHECKSUMS = {
    "C2_1920x1080_16spp_256frames": (1920, 1080, 16, 256, 0, ("default", 1)),
    "C3_3840x2160_64spp_8frames": (3840, 2160, 64, 8, O.ACCUM_F32 | O.SHADE_CHECKER, ("bunch", 20, 3, 3)),
    "C4_1920x1080_4spp_16frames": (1920, 1080, 4, 16, 0, ("bunch", 1024, 16, 4)),
    "C5_7680x4320_1spp_first16frames": (7680, 4320, 1, 16, O.ACCUM_F32 | O.COMBINE_SUM, ("default", 1)),
}


def scene(spec):
    return host.scene_default(spec[1]) if spec[0] == "default" else host.scene_random_bunch(spec[1], spec[2], spec[3])


def run(W, H, spp, frames, flags, spec):
    spheres, mats = scene(spec)
    cam = host.camera_raw(W, H, spp=spp, frames=frames)
    tb = host.time_table(frames)
    img = O.OracleImage(W, H, flags)
    O.render(img, spheres, mats, cam, 0, frames)
    return spheres, mats, cam, tb, img


def digest(img):
    d = {"rgba8_sha256": hashlib.sha256(np.ascontiguousarray(img.view_rgba8()).tobytes()).hexdigest(),
         "rng_sha256": hashlib.sha256(img.rng.tobytes()).hexdigest(), "counters": img.counters_dict()}
    if img.accum is not None:
        d["accum_sha256"] = hashlib.sha256(img.accum.tobytes()).hexdigest()
    return d


def main():
    full = "--full" in sys.argv
    path = os.path.join(HERE, "checksums.json")
    sums = json.load(open(path)) if os.path.exists(path) else {}
    for name, (W, H, spp, frames, flags, spec) in CASES.items():
        spheres, mats, cam, tb, img = run(W, H, spp, frames, flags, spec)
        out = dict(width=W, height=H, spp=spp, frames=frames, flags=flags, spheres=spheres.view(np.uint8), materials=mats.view(np.uint8),
                   camera=np.frombuffer(cam.tobytes(), np.uint8), time_bits=tb, rgba8=img.view_rgba8(), rng=img.rng,
                   counters=img.counters)
        if img.accum is not None:
            out["accum"] = img.accum
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "written")
    for name, (W, H, spp, frames, flags, spec) in CHECKSUMS.items():
        _s, _m, _c, _t, img = run(W, H, spp, frames, flags, spec)
        sums[name] = digest(img)
        print(name, sums[name]["rgba8_sha256"][:16])
    for name, (W, H, spp, frames, flags, spec) in (FULL_CHECKSUMS.items() if full else ()):
        _s, _m, _c, _t, img = run(W, H, spp, frames, flags, spec)
        sums[name] = digest(img)
        print(name, sums[name]["rgba8_sha256"][:16], flush=True)
    with open(path, "w") as f:
        json.dump(sums, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
