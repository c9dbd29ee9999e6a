"""CPU tests of the oracle itself: quirks of the reference it must reproduce and
size-independent properties that also guard the HIP path's tests."""
import numpy as np
import pytest

import oracle_lib as O
import scenes


def _render(W, H, spp, frames, flags=0, scene=None, rank=0, world=1, n=None):
    spheres, mats = scene or scenes.default_scene(1)
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames)
    img = O.OracleImage(W, H, flags)
    O.render(img, spheres, mats, cam, 0, frames if n is None else n, rank=rank, world=world)
    return img


def test_layout_sizes_match_the_abi_header():
    lib = O.load()
    assert [lib.rto_sizeof(i) for i in range(4)] == [32, 32, 160, 96]
    assert O.SPHERE_DTYPE.itemsize == 32 and O.MATERIAL_DTYPE.itemsize == 32 and O.CAMERA_DTYPE.itemsize == 160
    assert O.CAMERA_DTYPE.fields["pixel_to_world"][1] == 64 and O.CAMERA_DTYPE.fields["current_frame"][1] == 156
    assert O.SPHERE_DTYPE.fields["radius"][1] == 16 and O.SPHERE_DTYPE.fields["material_index"][1] == 20


def test_floor_dispatch_leaves_ragged_rows_untouched():
    # raytracing.rs:192: 640x360 -> 40x22 groups; rows 352..359 are never written (stay 0,0,0,0)
    img = _render(640, 360, 1, 1)
    assert not img.rgba8[352:].any()
    assert (img.rgba8[:352, :, 3] == 255).all()
    assert not img.rng[352:].any() and img.rng[:352].all()
    img = _render(70, 50, 1, 1)
    assert not img.rgba8[48:].any() and not img.rgba8[:, 64:].any()
    full = _render(70, 50, 1, 1, flags=O.DISPATCH_FULL)
    assert (full.rgba8[..., 3] == 255).all()
    assert np.array_equal(full.rgba8[:48, :64], img.rgba8[:48, :64])


def test_counter_identities():
    img = _render(96, 64, 3, 2)
    c = img.counters_dict()
    assert c["P"] == 96 * 64 * 2 and c["N"] == c["P"] * 3
    assert c["R"] == c["H"] + c["M"] and c["S"] == c["H"] and c["S"] == c["Sm"] + c["Sd"]
    assert c["T"] == c["R"] * 4
    assert c["rand_calls"] == c["P"] + 2 * c["N"] + 3 * c["J"]
    assert c["R"] <= 11 * c["N"]  # for depth = 0; depth <= 10 (wgsl:131)
    # rejection sampling accepts with probability pi/6 (wgsl:281-290)
    assert abs(c["Sd"] / c["J"] - np.pi / 6) < 0.01


def test_frames_past_frames_to_render_are_noops():
    a = _render(48, 32, 1, 3, n=3)
    b = _render(48, 32, 1, 3, n=7)
    assert np.array_equal(a.rgba8, b.rgba8) and np.array_equal(a.rng, b.rng)
    assert a.counters_dict() == b.counters_dict()


def test_seed_depends_on_pixel_and_time_only():
    # same pixel, same time bits -> same stream whatever the frame counter says (wgsl:255-261)
    spheres, mats = scenes.default_scene(1)
    cam0 = scenes.camera_raw(32, 32, spp=1, frames=100, current_frame=0)
    cam5 = scenes.camera_raw(32, 32, spp=1, frames=100, current_frame=5)
    a, b = O.OracleImage(32, 32, 0), O.OracleImage(32, 32, 0)
    O.compute(a, spheres, mats, cam0, 0x3C23D70A)
    O.compute(b, spheres, mats, cam5, 0x3C23D70A)
    assert np.array_equal(a.rng, b.rng)


@pytest.mark.parametrize("world", [2, 5])
@pytest.mark.parametrize("flags", [0, O.ACCUM_F32])
def test_tile_partition_is_disjoint_and_sums_to_the_whole(world, flags):
    whole = _render(80, 48, 2, 3, flags=flags)
    tot8 = np.zeros_like(whole.rgba8, dtype=np.int64)
    totf = np.zeros((48, 80, 4), np.float64)
    cnt = {k: 0 for k in O.COUNTER_NAMES}
    for r in range(world):
        part = _render(80, 48, 2, 3, flags=flags, rank=r, world=world)
        tot8 += part.rgba8
        if flags & O.ACCUM_F32:
            totf += part.accum
        for k, v in part.counters_dict().items():
            cnt[k] += v
    if flags & O.ACCUM_F32:
        assert np.array_equal(totf.astype(np.float32).view(np.uint32), whole.accum.view(np.uint32))
    else:
        assert np.array_equal(tot8.astype(np.uint8), whole.rgba8)
    assert cnt == whole.counters_dict()


def test_frame_split_sum_equals_running_average_within_rounding():
    frames = 16
    avg = _render(64, 48, 1, frames, flags=O.ACCUM_F32)
    summed = _render(64, 48, 1, frames, flags=O.ACCUM_F32 | O.COMBINE_SUM)
    mean = summed.accum / np.float32(frames)
    assert np.abs(mean - avg.accum).max() < 2e-6
    assert (summed.accum[..., 3] == frames).all()


def test_rgba8_running_average_requantises_every_frame():
    # the 8-bit recurrence differs from quantising the float average once (a17)
    frames = 32
    q = _render(64, 48, 1, frames)
    f = _render(64, 48, 1, frames, flags=O.ACCUM_F32)
    diff = np.abs(q.rgba8.astype(int) - f.view_rgba8().astype(int))
    assert 1 <= diff.max() <= 8  # a few LSB of drift from 32 re-quantisations
    psnr = 10 * np.log10(255.0**2 / np.mean(diff.astype(np.float64) ** 2))
    assert psnr > 40.0


def test_checker_gate_equals_sign_of_sin_times_cos():
    # RT_SHADE_CHECKER's floor-parity gate == (sin(4 pi nx) * cos(4 pi ny) > 0), the dead terms at wgsl:157-158
    rs = np.random.RandomState(0)
    n = rs.uniform(-1, 1, (200000, 2)).astype(np.float32)
    fx = np.floor(np.float32(4) * n[:, 0])
    fy = np.floor(np.float32(4) * n[:, 1] + np.float32(0.5))
    h = (fx + fy) * np.float32(0.5)
    gate = h == np.floor(h)
    b = np.sin(4 * np.pi * n[:, 0].astype(np.float64))
    c = np.cos(4 * np.pi * n[:, 1].astype(np.float64))
    clear = (np.abs(b) > 1e-4) & (np.abs(c) > 1e-4)
    assert np.array_equal(gate[clear], (b * c > 0)[clear])


def test_checker_scene_has_dark_sky_and_glow():
    img = _render(96, 64, 8, 2, flags=O.SHADE_CHECKER | O.ACCUM_F32, scene=scenes.random_bunch(20, 3, 3))
    c = img.counters_dict()
    assert c["M"] > 0 and c["H"] > 0
    rgb = img.accum[..., :3]
    assert rgb.min() >= 0.0 and rgb.max() > 0.2
    assert (rgb.sum(-1) == 0).mean() > 0.05  # sky off: some pixels are exactly black


def test_bad_material_index_is_rejected():
    spheres, mats = scenes.default_scene(1)
    spheres = spheres.copy()
    spheres["material_index"][2] = 3
    img = O.OracleImage(16, 16, 0)
    with pytest.raises(ValueError):
        O.compute(img, spheres, mats, scenes.camera_raw(16, 16), 0x3C23D70A)
