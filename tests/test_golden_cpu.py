"""The oracle must keep reproducing the committed golden vectors (tests/golden/make_golden.py)."""
import hashlib

import numpy as np
import pytest

import golden_lib as G
import oracle_lib as O


@pytest.mark.parametrize("name", G.NAMES)
def test_oracle_reproduces_golden(name):
    g = G.load(name)
    img = O.OracleImage(g["width"], g["height"], g["flags"])
    cam = np.array(g["camera"], copy=True)
    for f in range(g["frames"]):
        O.compute(img, g["spheres"], g["materials"], cam, int(g["time_bits"][f]))
        cam["current_frame"] += 1
    assert np.array_equal(img.view_rgba8(), g["rgba8"])
    assert np.array_equal(img.rng, g["rng"])
    assert np.array_equal(img.counters, g["counters"])
    if "accum" in g:
        assert np.array_equal(img.accum.view(np.uint32), g["accum"].view(np.uint32))


def test_c1_complete_checksum():
    """BASELINE config 1 in full (640x360, 1 spp x 64 frames, RGBA8 recurrence)."""
    from raytracer_amd import host

    want = G.checksums()["C1_640x360_1spp_64frames"]
    spheres, mats = host.scene_default(1)
    cam = host.camera_raw(640, 360, spp=1, frames=64)
    img = O.OracleImage(640, 360, 0)
    O.render(img, spheres, mats, cam, 0, 64)
    assert hashlib.sha256(img.rgba8.tobytes()).hexdigest() == want["rgba8_sha256"]
    assert hashlib.sha256(img.rng.tobytes()).hexdigest() == want["rng_sha256"]
    assert img.counters_dict() == want["counters"]
