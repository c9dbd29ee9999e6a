"""Double-entry check of the oracle's FLOAT path (CPU): oracle/rt_oracle.c against tests/np_oracle.py, a second
restatement of raytrace_kernal.wgsl:85-231 written independently in numpy float32 (one explicit binary32 operation at
a time, vectorised over pixels). Bit for bit: RGBA8 image, float image, every pixel's final RNG state, rand() count.

The reference holds no vectors and cannot run here (parity unpinned); two transcriptions that agree is the only
mitigation available. It pins how the shader text was READ under DESIGN.md's arithmetic contract, not what a wgpu
driver would compute.
"""
import numpy as np
import pytest

import np_oracle as NP
import oracle_lib as O
import scenes

CASES = [
    # name, W, H, spp, frames, flags, scene, camera kwargs
    ("default_rgba8_ragged_floor16", 70, 50, 2, 3, 0, "default", {}),
    ("default_f32_full_dispatch_ragged", 37, 21, 3, 2, O.ACCUM_F32 | O.DISPATCH_FULL, "default", {}),
    ("checker_bunch20_f32", 64, 48, 3, 2, O.ACCUM_F32 | O.SHADE_CHECKER, "bunch20", {}),
    ("checker_bunch20_rgba8", 48, 32, 2, 2, O.SHADE_CHECKER, "bunch20", {}),
    ("moved_camera_sum", 48, 48, 2, 3, O.ACCUM_F32 | O.COMBINE_SUM, "default",
     dict(pos=(0.4, 0.3, 0.8), look=(-0.2, -0.1, -1.0), fov=0.35)),
    ("inside_a_mirror_sphere_deep_paths", 32, 32, 2, 2, 0, "inside", {}),
    ("bunch40_mixed_radii", 48, 32, 2, 2, 0, "mixed", {}),
]


def _scene(name):
    if name == "default":
        return scenes.default_scene(1)
    if name == "bunch20":
        return scenes.random_bunch(20, 3, 3)
    if name == "inside":  # camera inside sphere 0 (even index: mirror): paths run to MAX_RAY_DEPTH without a miss
        mats = scenes.make_materials([(0.9, 0.2, 0.1, 1.0), (0.1, 0.8, 0.3, 1.0)])
        return scenes.make_spheres([(0.0, 0.0, 0.0), (0.5, 0.2, -1.5), (-0.7, -0.3, -1.2)], [4.0, 0.6, 0.45], [0, 1, 1]), mats
    if name == "mixed":
        rs = np.random.RandomState(7)
        n = 40
        centers = (rs.random_sample((n, 3)) * 2 - 1) * np.array([4, 3, 3]) + np.array([0, 0, -6])
        radii = rs.choice([0.25, 0.5, 0.3, 0.77, 1.1], size=n)  # powers of two and not
        return scenes.make_spheres(centers, radii, rs.randint(0, 4, size=n)), scenes.make_materials(rs.random_sample((4, 4)))
    raise KeyError(name)


@pytest.mark.parametrize("name,W,H,spp,frames,flags,scene,camkw", CASES, ids=[c[0] for c in CASES])
def test_numpy_restatement_equals_the_c_oracle(name, W, H, spp, frames, flags, scene, camkw):
    spheres, mats = _scene(scene)
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames, **camkw)
    img = O.OracleImage(W, H, flags)
    O.render(img, spheres, mats, cam, 0, frames)
    rgba8, f32, rng, calls = NP.render(W, H, flags, spheres, mats, cam, frames)
    assert rng.any() and np.array_equal(rng, img.rng), "final RNG states differ"
    assert calls == img.counters_dict()["rand_calls"]
    if flags & O.ACCUM_F32:
        assert np.array_equal(f32.view(np.uint32), img.accum.view(np.uint32)), "float images differ"
        assert np.array_equal(rgba8, img.view_rgba8())
    else:
        assert np.array_equal(rgba8, img.rgba8), "RGBA8 images differ"
    assert rgba8[..., :3].any()


def test_frames_past_the_end_and_time_schedule():
    """frames_to_render = 2 but 4 redraws: the last two are no-ops (wgsl:76); time keeps advancing (time.rs:72-77)."""
    W, H = 32, 32
    spheres, mats = scenes.default_scene(1)
    cam = scenes.camera_raw(W, H, spp=1, frames=2)
    img = O.OracleImage(W, H, 0)
    O.render(img, spheres, mats, cam, 0, 4)
    rgba8, _f32, rng, _calls = NP.render(W, H, 0, spheres, mats, cam, 4)
    assert np.array_equal(rgba8, img.rgba8) and np.array_equal(rng, img.rng)
