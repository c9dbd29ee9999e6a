"""GPU parity: the HIP path through the C ABI vs the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star):
  RT_MATH_STRICT  bit-exact: RGBA8 image, float4 image, every pixel's final RNG state after
                  every frame, and all event counters.
  RT_MATH_FAST    the RNG function is the same integer code (seed + first draws checked through
                  pixels whose path has no float-dependent decision... see test_fast_*), and the
                  converged image is within max-abs <= 3/255 and PSNR >= 40 dB of the strict one.
"""
import numpy as np
import pytest

import oracle_lib as O
import scenes

pytestmark = pytest.mark.gpu


def _rt():
    import raytracer_amd as R

    return R


def _run_gpu(R, W, H, flags, spheres, mats, cam, n_frames, per_frame_check=None, batched=False, rank=0, world=1,
             counters=False):
    rt = R.Raytracer(W, H, flags)
    rt.set_scene(spheres, mats)
    rt.set_partition(rank, world)
    rt.enable_rng_capture(True)
    if counters:
        rt.enable_counters(True)
    tb = O.time_bits_table(n_frames)
    cam = np.array(cam, copy=True)
    if batched:
        rt.set_camera(cam)
        rt.render_frames(tb)
    else:
        for f in range(n_frames):
            rt.set_time(int(tb[f]))  # App::update (app.rs:94)
            rt.set_camera(cam)
            rt.compute()  # App::render (app.rs:106-108)
            if per_frame_check is not None:
                per_frame_check(f, rt)
            cam["current_frame"] += 1  # Camera::increase_frame (camera.rs:253)
    rt.sync()
    return rt


CASES = [
    # name, W, H, spp, frames, flags, scene
    ("c1_small_rgba8", 64, 40, 1, 4, 0, "default"),
    ("ragged_floor16", 70, 50, 2, 3, 0, "default"),
    ("full_dispatch_ragged", 70, 50, 2, 2, O.DISPATCH_FULL, "default"),
    ("f32_accum", 64, 48, 3, 3, O.ACCUM_F32, "default"),
    ("f32_sum", 64, 48, 2, 3, O.ACCUM_F32 | O.COMBINE_SUM, "default"),
    ("checker_bunch20", 96, 64, 4, 2, O.SHADE_CHECKER | O.ACCUM_F32, "bunch20"),
    ("bunch1024", 48, 32, 2, 2, 0, "bunch1024"),
    ("one_sphere", 32, 32, 2, 2, 0, "one"),
]


def _scene(name):
    if name == "default":
        return scenes.default_scene(1)
    if name == "bunch20":
        return scenes.random_bunch(20, 3, 3)
    if name == "bunch1024":
        return scenes.random_bunch(1024, 16, 4)
    if name == "one":
        s, m = scenes.default_scene(2)
        return s[1:2], m
    raise KeyError(name)


@pytest.mark.parametrize("name,W,H,spp,frames,flags,scene", CASES, ids=[c[0] for c in CASES])
def test_strict_bit_exact_per_frame(name, W, H, spp, frames, flags, scene):
    R = _rt()
    spheres, mats = _scene(scene)
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames)
    img = O.OracleImage(W, H, flags)
    ocam = np.array(cam, copy=True)
    tb = O.time_bits_table(frames)

    def check(f, rt):
        O.compute(img, spheres, mats, ocam, int(tb[f]))
        ocam["current_frame"] += 1
        assert np.array_equal(rt.read_rng_state(), img.rng), f"{name}: RNG state differs at frame {f}"
        if flags & O.ACCUM_F32:
            got = rt.read_accum_f32()
            assert np.array_equal(got.view(np.uint32), img.accum.view(np.uint32)), f"{name}: float image differs at frame {f}"
        assert np.array_equal(rt.read_rgba8(), img.view_rgba8()), f"{name}: RGBA8 image differs at frame {f}"

    rt = _run_gpu(R, W, H, flags | R.RT_MATH_STRICT, spheres, mats, cam, frames, per_frame_check=check, counters=True)
    assert rt.counters() == img.counters_dict()
    rt.close()


@pytest.mark.parametrize("flags", [0, O.ACCUM_F32], ids=["rgba8", "f32"])
def test_batched_launch_equals_frame_loop(flags):
    """rt_render_frames (one launch for n frames) == n x (set_time, compute, current_frame += 1)."""
    R = _rt()
    W, H, spp, frames = 64, 48, 2, 6
    spheres, mats = scenes.default_scene(1)
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames)
    a = _run_gpu(R, W, H, flags, spheres, mats, cam, frames, batched=False)
    b = _run_gpu(R, W, H, flags, spheres, mats, cam, frames, batched=True)
    assert np.array_equal(a.read_rgba8(), b.read_rgba8())
    assert np.array_equal(a.read_rng_state(), b.read_rng_state())
    if flags & O.ACCUM_F32:
        assert np.array_equal(a.read_accum_f32().view(np.uint32), b.read_accum_f32().view(np.uint32))
    # and against the oracle's App loop
    img = O.OracleImage(W, H, flags)
    O.render(img, spheres, mats, cam, 0, frames)
    assert np.array_equal(b.read_rgba8(), img.view_rgba8())
    assert np.array_equal(b.read_rng_state(), img.rng)
    a.close(); b.close()


def test_frames_past_the_end_are_noops():
    """current_frame >= frames_to_render: compute does nothing (raytrace_kernal.wgsl:76)."""
    R = _rt()
    W, H = 32, 32
    spheres, mats = scenes.default_scene(1)
    cam = scenes.camera_raw(W, H, spp=1, frames=2)
    rt = _run_gpu(R, W, H, 0, spheres, mats, cam, 5)  # 5 redraws, only 2 render
    img = O.OracleImage(W, H, 0)
    O.render(img, spheres, mats, cam, 0, 5)
    assert np.array_equal(rt.read_rgba8(), img.rgba8)
    rt2 = _run_gpu(R, W, H, 0, spheres, mats, cam, 5, batched=True)
    assert np.array_equal(rt2.read_rgba8(), img.rgba8)
    rt.close(); rt2.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_tile_partition_sums_to_the_whole(world):
    """Multi-GPU tile split on one device: the per-rank images are disjoint and sum to the 1-GPU image."""
    R = _rt()
    W, H, spp, frames = 80, 64, 2, 3
    flags = O.ACCUM_F32
    spheres, mats = scenes.default_scene(1)
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames)
    whole = _run_gpu(R, W, H, flags, spheres, mats, cam, frames, batched=True)
    ref = whole.read_accum_f32()
    total = np.zeros_like(ref)
    covered = np.zeros((H, W), np.int32)
    for r in range(world):
        part = _run_gpu(R, W, H, flags, spheres, mats, cam, frames, batched=True, rank=r, world=world)
        p = part.read_accum_f32()
        covered += (p[..., 3] != 0).astype(np.int32)
        total += p
        # the oracle's own partition agrees tile for tile
        img = O.OracleImage(W, H, flags)
        O.render(img, spheres, mats, cam, 0, frames, rank=r, world=world)
        assert np.array_equal(p.view(np.uint32), img.accum.view(np.uint32))
        part.close()
    assert covered.max() == 1
    assert np.array_equal(total.view(np.uint32), ref.view(np.uint32))
    whole.close()


def _psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10.0 * np.log10(1.0 / mse)


@pytest.mark.parametrize("scene,shade", [("default", 0), ("bunch20", O.SHADE_CHECKER)], ids=["head", "checker"])
def test_fast_math_within_tolerance(scene, shade):
    """RT_MATH_FAST vs the strict oracle, converged float image: max-abs <= 3/255, PSNR >= 40 dB
    (the tolerance SURVEY.md §8a/a17 proposes), and identical final RNG state for >= 99 % of pixels
    (a differing state means a float-borderline hit/miss decision flipped, not a different RNG)."""
    R = _rt()
    W, H, spp, frames = 128, 96, 16, 8
    flags = O.ACCUM_F32 | shade
    spheres, mats = _scene(scene)
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames)
    fast = _run_gpu(R, W, H, flags | R.RT_MATH_FAST, spheres, mats, cam, frames, batched=True)
    img = O.OracleImage(W, H, flags)
    O.render(img, spheres, mats, cam, 0, frames)
    got, ref = np.clip(fast.read_accum_f32()[..., :3], 0, 1), np.clip(img.accum[..., :3], 0, 1)
    same = np.mean(fast.read_rng_state() == img.rng)
    max_abs = np.abs(got - ref).max()
    psnr = _psnr(got, ref)
    print(f"fast-vs-strict[{scene}]: rng-identical {same:.5f}, max-abs {max_abs:.5f}, PSNR {psnr:.1f} dB")
    assert same >= 0.99
    assert psnr >= 40.0
    err = np.abs(got - ref).max(axis=-1)
    mask = fast.read_rng_state() == img.rng
    if scene == "default":
        # pixels whose RNG trajectory is identical made the same random decisions; a pixel whose
        # trajectory diverged is a different, equally valid Monte-Carlo sample
        assert err[mask].max() <= 3.0 / 255.0
    else:
        # mirror-to-mirror chains between spheres are chaotic (a 1-ulp change grows ~10x per
        # bounce), so deep paths decorrelate without touching the RNG: bound the outlier fraction
        assert np.mean(err > 3.0 / 255.0) <= 0.02
    fast.close()


def test_fast_rng_function_is_bit_exact():
    """A scene with no spheres in view: every path is a miss after the two jitter draws, so the final
    RNG state depends on integer code only — it must be bit-exact in fast mode too."""
    R = _rt()
    W, H, spp, frames = 64, 48, 5, 3
    spheres, mats = scenes.default_scene(1)
    spheres = spheres[1:2].copy()
    spheres["center"] = (0.0, 0.0, 50.0)  # behind the camera
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames)
    fast = _run_gpu(R, W, H, R.RT_MATH_FAST, spheres, mats, cam, frames, batched=True)
    img = O.OracleImage(W, H, 0)
    O.render(img, spheres, mats, cam, 0, frames)
    assert np.array_equal(fast.read_rng_state(), img.rng)
    assert np.abs(fast.read_rgba8().astype(int) - img.rgba8.astype(int)).max() <= 1
    fast.close()


def test_errors_are_codes_not_crashes():
    R = _rt()
    with pytest.raises(R.RtError):
        R.Raytracer(0, 10)
    with pytest.raises(R.RtError):
        R.Raytracer(16, 16, R.RT_COMBINE_SUM)  # needs F32
    rt = R.Raytracer(16, 16)
    with pytest.raises(R.RtError):
        rt.compute()  # nothing set
    spheres, mats = scenes.default_scene(1)
    bad = spheres.copy()
    bad["material_index"][0] = 7
    with pytest.raises(R.RtError):
        rt.set_scene(bad, mats)
    cam = scenes.camera_raw(16, 16, spp=0)
    with pytest.raises(R.RtError):
        rt.set_camera(cam)
    rt.close()


def test_torch_owned_image_and_single_rank_rccl_reduce():
    """rt_bind_accum: the kernel accumulates into a torch tensor, and that same tensor goes through a
    (1-rank) RCCL reduce — the memory path bench.py --gpus N uses."""
    import os

    import torch
    import torch.distributed as dist

    R = _rt()
    from raytracer_amd import dist as rdist

    W, H, spp, frames = 64, 48, 2, 3
    flags = O.ACCUM_F32
    spheres, mats = scenes.default_scene(1)
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames)
    rt = R.Raytracer(W, H, flags)
    rt.set_scene(spheres, mats)
    img = rdist.DeviceImage(rt, "cuda:0")
    rt.set_camera(cam)
    rt.render_frames(O.time_bits_table(frames))
    rt.sync()
    started = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        started = True
    ref = O.OracleImage(W, H, flags)
    O.render(ref, spheres, mats, cam, 0, frames)
    try:
        rdist.combine_sum(img.tensor, dst=0)
        torch.cuda.synchronize()
        # the tile split's other combine: pack my tiles, (1-rank) RCCL gather, unpack - on the image itself
        assert rdist.TileGather.usable(W, H)
        rdist.TileGather(img.tensor, 0, 1, dst=0).run()
        torch.cuda.synchronize()
        # ... and its pack / unpack on the device for three ranks: rank r's packed tiles, put back where rank 0 would
        # put them, rebuild the picture from three images that each hold a third of the tiles
        whole = img.tensor.clone()
        rebuilt = torch.zeros_like(whole)
        for r in range(3):
            part = whole * torch.from_numpy(rdist.tile_owner_mask(W, H, 3, r)).to(whole.device)[..., None]
            g = rdist.TileGather(part, r, 3, dst=r)
            n = g.mine[0].numel()
            g.send[:n] = g.tiles[g.mine[0], g.mine[1]]
            dstg = rdist.TileGather(rebuilt, 0, 3, dst=0)
            ty, tx = dstg.theirs[r]
            dstg.tiles[ty, tx] = g.send[: ty.numel()]
        torch.cuda.synchronize()
        assert torch.equal(rebuilt.view(torch.int32), whole.view(torch.int32))
    finally:
        if started:
            dist.destroy_process_group()
    assert np.array_equal(img.tensor.cpu().numpy().view(np.uint32), ref.accum.view(np.uint32))
    assert np.array_equal(rt.read_accum_f32().view(np.uint32), ref.accum.view(np.uint32))
    rt.bind_accum(0, 0)  # back to the context's own image before the tensor dies
    rt.close()


@pytest.mark.parametrize("flags", [0, O.ACCUM_F32], ids=["rgba8", "f32"])
def test_checkpoint_resume_is_bit_identical(flags):
    """Dump {image, current_frame, time bits} after 3 frames, resume in a fresh context for 3 more:
    equals 6 frames straight (SURVEY.md §5 checkpoint/resume, §8f-4)."""
    R = _rt()
    W, H, spp, frames = 64, 48, 2, 6
    spheres, mats = scenes.default_scene(1)
    tb = O.time_bits_table(frames)
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames)
    straight = _run_gpu(R, W, H, flags, spheres, mats, cam, frames, batched=True)
    a = R.Raytracer(W, H, flags); a.set_scene(spheres, mats); a.set_camera(cam); a.render_frames(tb[:3])
    ckpt = a.read_accum_f32() if flags & O.ACCUM_F32 else a.read_rgba8()
    a.close()
    b = R.Raytracer(W, H, flags); b.set_scene(spheres, mats); b.enable_rng_capture(True)
    b.write_image(ckpt)
    cam2 = np.array(cam, copy=True); cam2["current_frame"] = 3
    b.set_camera(cam2); b.render_frames(tb[3:])
    assert np.array_equal(b.read_rgba8(), straight.read_rgba8())
    assert np.array_equal(b.read_rng_state(), straight.read_rng_state())
    with pytest.raises(R.RtError):
        b.write_image(ckpt[:10])
    b.close(); straight.close()


def test_degenerate_sizes_and_long_batches():
    """Images smaller than one 16x16 workgroup render nothing under the reference's floor dispatch; a
    batch longer than the initial frame-table capacity (256) grows it."""
    R = _rt()
    spheres, mats = scenes.default_scene(1)
    rt = R.Raytracer(15, 40, 0); rt.set_scene(spheres, mats)
    rt.set_camera(scenes.camera_raw(15, 40, spp=1, frames=2)); rt.set_time(0x3C23D70A); rt.compute()
    assert not rt.read_rgba8().any()
    rt.close()
    W, H, frames = 32, 16, 300
    cam = scenes.camera_raw(W, H, spp=1, frames=frames)
    rt = _run_gpu(R, W, H, 0, spheres, mats, cam, frames, batched=True)
    img = O.OracleImage(W, H, 0)
    O.render(img, spheres, mats, cam, 0, frames)
    assert np.array_equal(rt.read_rgba8(), img.rgba8) and np.array_equal(rt.read_rng_state(), img.rng)
    rt.close()


def test_cost_ordered_tile_queue_does_not_change_results(monkeypatch):
    """With few persistent waves the queue hands tiles out in the previous launch's cost order; the
    image and RNG states must not depend on it (launches 2.. run ordered)."""
    R = _rt()
    monkeypatch.setenv("RT_WAVES_PER_CU", "1")  # 256 waves for 1 800 tiles -> ordering is active
    W, H, spp, frames = 480, 240, 2, 4
    spheres, mats = scenes.default_scene(1)
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames)
    img = O.OracleImage(W, H, 0)
    ocam = np.array(cam, copy=True)
    tb = O.time_bits_table(frames)

    def check(f, rt):
        O.compute(img, spheres, mats, ocam, int(tb[f]))
        ocam["current_frame"] += 1
        assert np.array_equal(rt.read_rng_state(), img.rng), f"frame {f}"
        assert np.array_equal(rt.read_rgba8(), img.rgba8), f"frame {f}"

    rt = _run_gpu(R, W, H, 0, spheres, mats, cam, frames, per_frame_check=check)
    rt.close()
    monkeypatch.setenv("RT_TILE_ORDER", "0")
    rt2 = _run_gpu(R, W, H, 0, spheres, mats, cam, frames)
    assert np.array_equal(rt2.read_rgba8(), img.rgba8)
    rt2.close()


def _strict_equals_oracle(R, W, H, spp, frames, flags, spheres, mats, cam, counters):
    """Strict launch(es) against the oracle: image, RNG states (and counters when counting, which also
    selects the kernel variant without the behind-the-ray cull)."""
    img = O.OracleImage(W, H, flags)
    O.render(img, spheres, mats, cam, 0, frames)
    rt = _run_gpu(R, W, H, flags | R.RT_MATH_STRICT, spheres, mats, cam, frames, counters=counters)
    assert np.array_equal(rt.read_rng_state(), img.rng)
    if flags & O.ACCUM_F32:
        assert np.array_equal(rt.read_accum_f32().view(np.uint32), img.accum.view(np.uint32))
    assert np.array_equal(rt.read_rgba8(), img.view_rgba8())
    if counters:
        assert rt.counters() == img.counters_dict()
    rt.close()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 31, 32, 33, 35, 64, 65, 127, 128, 129, 130])
def test_sphere_counts_around_the_pair_and_group_sizes(n):
    """Phase 1 handles spheres four at a time (padding spheres that never hit) in groups of 32, and
    switches the behind cull off above 128 spheres: every remainder must give the oracle's image."""
    R = _rt()
    spheres, mats = scenes.random_bunch(n, 3, 100 + n)
    cam = scenes.camera_raw(64, 48, spp=2, frames=2)
    for counters in (False, True):
        _strict_equals_oracle(R, 64, 48, 2, 2, O.ACCUM_F32, spheres, mats, cam, counters)


def _edge_scene(name):
    mats = scenes.make_materials(np.random.RandomState(5).random_sample((3, 4)).astype(np.float32))
    if name == "camera_inside_a_sphere":      # origin inside: no cull, hits from the inside, second roots
        return scenes.make_spheres([(0, 0, 0), (0.5, 0, -2), (-0.7, 0.2, -2.5)], [6.0, 0.5, 0.75], [0, 1, 2]), mats, {}
    if name == "coincident_and_touching":     # equal roots, tangent spheres, one sphere inside another
        return scenes.make_spheres([(0, 0, -3), (0, 0, -3), (1.0, 0, -3), (0, 0, -3), (0, -1.5, -3)],
                                   [0.5, 0.5, 0.5, 0.25, 1.0], [0, 1, 2, 0, 1]), mats, {}
    if name == "tiny_and_huge_radii":
        return scenes.make_spheres([(0, 0, -2), (0.3, 0.1, -1.5), (0, -1000.5, -2), (0, 0, -50)],
                                   [1e-3, 0.125, 1000.0, 30.0], [0, 1, 2, 0]), mats, {}
    if name == "far_from_the_origin":         # |c|^2 ~ 3e6: the margin is relative to |o|^2 + |c|^2 + r^2
        c0 = np.array([1000.0, 1000.0, -1000.0])
        s, _ = scenes.random_bunch(9, 3, 7)
        s["center"] += c0.astype(np.float32)
        s["padding"] = s["center"][:, 2].copy().view(np.uint32)
        return s, mats, {"pos": tuple(c0)}
    if name == "outside_the_tame_range":      # |c|^2 > 2^40: the filter is skipped, every sphere is tested exactly
        return scenes.make_spheres([(0, 0, -3), (2.0e6, 0, -3), (0.6, 0.1, -2)], [1.0, 1.0e6, 0.25], [0, 1, 2]), mats, {}
    if name == "degenerate_radius_zero":
        return scenes.make_spheres([(0, 0, -3), (0.2, 0.2, -2)], [0.0, 0.5], [0, 1]), mats, {}
    raise KeyError(name)


@pytest.mark.parametrize("name", ["camera_inside_a_sphere", "coincident_and_touching", "tiny_and_huge_radii",
                                  "far_from_the_origin", "outside_the_tame_range", "degenerate_radius_zero"])
def test_phase1_filter_edge_scenes(name):
    """Scenes that stress the phase-1 filter (margin, behind cull, tame range): results stay the oracle's."""
    R = _rt()
    spheres, mats, camkw = _edge_scene(name)
    cam = scenes.camera_raw(64, 48, spp=3, frames=2, **camkw)
    for counters in (False, True):
        _strict_equals_oracle(R, 64, 48, 3, 2, O.ACCUM_F32, spheres, mats, cam, counters)


@pytest.mark.parametrize("radius,height,ground_index", [(100.0, 0.5, 0), (100.0, 0.02, 0), (1000.0, 0.5, 0), (30000.0, 1.0, 0),
                                                        (100.0, 0.5, 1), (3.0, 0.01, 0)],
                         ids=["r100", "r100_low_camera", "r1000", "r30000", "r100_diffuse", "r3_skimming"])
def test_own_sphere_drop_on_big_spheres_at_grazing_angles(radius, height, ground_index):
    """Bounce rays that leave a big sphere almost tangentially: the own-sphere drop (one inequality in hb, a and
    r^2 - |o - c|^2) decides per ray whether the exact test of the sphere the ray starts on can be left out. A low
    camera over a mirror (or diffuse) ground makes most reflections graze; the image stays the oracle's, acne included."""
    R = _rt()
    mats = scenes.make_materials(np.random.RandomState(9).random_sample((3, 4)).astype(np.float32))
    centers = [(0.0, -radius - height, 0.0), (-1.0, 0.0, -2.0), (1.0, 0.0, -2.0), (0.0, 0.0, -2.0), (0.0, 0.5, -6.0)]
    radii, idx = [radius, 0.5, 0.5, 0.5, 1.5], [0, 1, 2, 0, 1]
    if ground_index == 1:  # the ground second: odd index = diffuse (wgsl:163)
        centers[0], centers[1] = centers[1], centers[0]
        radii[0], radii[1] = radii[1], radii[0]
    spheres = scenes.make_spheres(centers, radii, idx)
    cam = scenes.camera_raw(256, 160, spp=4, frames=2)
    _strict_equals_oracle(R, 256, 160, 4, 2, O.ACCUM_F32, spheres, mats, cam, False)


@pytest.mark.parametrize("case", ["stacked", "nested_and_coincident", "bunch32_near", "bunch300_lists", "negative_and_zero_radii"])
def test_occlusion_drop_of_the_tile_masks_keeps_the_nearest_sphere(case, monkeypatch):
    """rt_tile_masks_kernel leaves a sphere out of a tile's candidates when another sphere is certainly hit in front
    of it by every primary ray of the tile. Spheres behind each other with small and zero depth gaps, one inside the
    other, the same sphere twice (the tie rule), dense bunches seen from close by, and the candidate lists of a
    300-sphere scene: images and RNG states stay the oracle's, and equal the render without tile masks."""
    R = _rt()
    rs = np.random.RandomState(77)
    mats = scenes.make_materials(rs.random_sample((3, 4)).astype(np.float32))
    camkw = {}
    if case == "stacked":
        spheres = scenes.make_spheres([(0, 0, -3), (0.02, 0, -3.05), (0, 0.1, -3.5), (0.3, 0, -6), (-0.5, 0.2, -2.2), (0, 0, -40)],
                                      [1.0, 1.0, 1.2, 2.0, 0.3, 30.0], [0, 1, 2, 0, 1, 2])
    elif case == "nested_and_coincident":
        spheres = scenes.make_spheres([(0, 0, -4), (0, 0, -4), (0, 0, -4), (0.5, 0, -4), (0, 0, -4.001)],
                                      [1.0, 1.0, 0.5, 1.0, 1.0], [1, 0, 2, 0, 1])
    elif case == "bunch32_near":
        spheres, mats = scenes.random_bunch(32, 3, 11)
        camkw = dict(pos=(0.5, 0.2, -1.0), look=(0.1, -0.1, -1.0), fov=0.45)
    elif case == "bunch300_lists":
        spheres, mats = scenes.random_bunch(300, 3, 12)
        camkw = dict(fov=0.3)
    else:
        spheres = scenes.make_spheres([(0, 0, -5), (0.2, 0, -9), (0, 0.5, -3), (-1, 0, -7)], [-2.0, 3.0, 0.0, -1.5], [0, 1, 2, 1])
    W, H = 320, 192
    cam = scenes.camera_raw(W, H, spp=2, frames=2, **camkw)
    _strict_equals_oracle(R, W, H, 2, 2, O.ACCUM_F32, spheres, mats, cam, False)
    monkeypatch.setenv("RT_SKY_TILES", "1")
    on = _render_state(R, W, H, O.ACCUM_F32 | 0x8, spheres, mats, cam, 2, False)
    monkeypatch.setenv("RT_SKY_TILES", "0")
    off = _render_state(R, W, H, O.ACCUM_F32 | 0x8, spheres, mats, cam, 2, False)
    assert np.array_equal(on[1], off[1]) and np.array_equal(on[2], off[2])  # fast arithmetic: masks change no bit either


@pytest.mark.parametrize("flags", [0, O.ACCUM_F32], ids=["rgba8", "f32"])
def test_checkpoint_file_resumes_bit_identically_and_rejects_mismatches(flags, tmp_path):
    """SURVEY 8f-4 through the host App: render 3 of 7 frames, save, load into a NEW App, finish; the image
    equals the uninterrupted render's and the oracle's. A checkpoint of another scene / size is refused."""
    R = _rt()
    from raytracer_amd import host

    W, H, spp, frames = 96, 64, 2, 7
    spheres, mats = scenes.default_scene(1)
    whole = host.App(W, H, flags, spheres, mats, spp=spp, frames=frames)
    whole.run(frames)
    want = whole.read_rgba8()
    whole.close()

    first = host.App(W, H, flags, spheres, mats, spp=spp, frames=frames)
    first.redraw(3)
    ck = str(tmp_path / "render.ckpt")
    first.save_checkpoint(ck)
    first.close()
    info = host.checkpoint_info(ck)
    assert info["current_frame"] == 3 and info["frames_to_render"] == frames and info["time_bits"] == int(O.time_bits_table(3)[2])

    second = host.App(W, H, flags, spheres, mats, spp=1, frames=1)  # spp / frame count come from the file
    second.load_checkpoint(ck)
    assert second.current_frame == 3 and second.time_bits == info["time_bits"]
    second.run(frames - 3)
    got = second.read_rgba8()
    second.close()
    assert np.array_equal(got, want)
    # the oracle on the host mirror's own camera bytes (its f32 matrices, not the tests' float64 derivation)
    cam = host.camera_raw(W, H, spp=spp, frames=frames)
    img2 = O.OracleImage(W, H, flags)
    O.render(img2, spheres, mats, cam, 0, frames)
    assert np.array_equal(got, img2.view_rgba8())

    other_scene, other_mats = scenes.random_bunch(4, 3, 9)
    for app in (host.App(W, H, flags, other_scene, other_mats, spp=spp, frames=frames),
                host.App(W + 16, H, flags, spheres, mats, spp=spp, frames=frames),
                host.App(W, H, flags ^ O.ACCUM_F32, spheres, mats, spp=spp, frames=frames)):
        with pytest.raises(Exception):
            app.load_checkpoint(ck)
        app.close()
    # the arithmetic mode may change across a resume
    fast = host.App(W, H, flags | R.RT_MATH_FAST, spheres, mats, spp=spp, frames=frames)
    fast.load_checkpoint(ck)
    fast.run(frames - 3)
    assert fast.current_frame == frames
    fast.close()


def test_headless_cli_checkpoint_round_trip(tmp_path):
    """rt_headless --stop-after / --save-checkpoint / --resume: the resumed PNG equals the uninterrupted one."""
    import os, subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "rt_headless")
    base = [exe, "--width", "128", "--height", "64", "--spp", "2", "--frames", "6", "--scene", "checkers", "--seed", "3"]
    a, b, ck = str(tmp_path / "a.png"), str(tmp_path / "b.png"), str(tmp_path / "r.ckpt")
    subprocess.run(base + ["--out", a], check=True, timeout=120, capture_output=True)
    subprocess.run(base + ["--stop-after", "2", "--save-checkpoint", ck], check=True, timeout=120, capture_output=True)
    out = subprocess.run(base + ["--resume", ck, "--out", b], check=True, timeout=120, capture_output=True, text=True).stdout
    assert "resumed" in out and "at frame 2 of 6" in out
    assert open(a, "rb").read() == open(b, "rb").read()


def _fuzz_scene(rs):
    """Random scenes that lean on the phase-1 filter's assumptions: mixed magnitudes, overlaps, a camera that
    may sit inside a sphere or far from the origin, zero and negative radii."""
    n = int(rs.choice([1, 2, 3, 4, 6, 9, 17, 40]))
    scale = float(rs.choice([0.05, 1.0, 1.0, 30.0, 3000.0]))
    centers = (rs.random_sample((n, 3)) * 2 - 1) * scale * np.array([1.0, 1.0, 1.0]) + np.array([0, 0, -1.5 * scale])
    radii = rs.random_sample(n) * scale * float(rs.choice([0.05, 0.5, 2.0]))
    kind = rs.randint(0, 6)
    if kind == 0:
        radii[rs.randint(n)] = 0.0
    elif kind == 1:
        radii[rs.randint(n)] *= -1.0          # the shader never checks the sign: r*r and 1/r take it as is
    elif kind == 2 and n > 1:
        centers[1] = centers[0]; radii[1] = radii[0]  # coincident spheres: equal roots, first index wins
    elif kind == 3:
        radii[rs.randint(n)] = 2.0 ** int(rs.randint(-6, 6))  # power-of-two radius: the multiply path
    mats = scenes.make_materials(rs.random_sample((3, 4)).astype(np.float32))
    spheres = scenes.make_spheres(centers.astype(np.float32), radii.astype(np.float32), rs.randint(0, 3, size=n))
    pos = tuple(((rs.random_sample(3) * 2 - 1) * scale * float(rs.choice([0.0, 0.3, 1.5]))).tolist())
    look = centers.mean(axis=0) - np.asarray(pos) + (rs.random_sample(3) * 2 - 1) * 0.2 * scale  # towards the scene
    if not np.isfinite(look).all() or np.linalg.norm(look) < 1e-6 * scale or abs(look[0]) + abs(look[2]) < 1e-6 * scale:
        look = np.array([0.1, 0.0, -1.0])
    return spheres, mats, dict(pos=pos, look=tuple(look.tolist()), fov=float(rs.choice([0.1, 0.25, 0.45])))


@pytest.mark.parametrize("seed", range(24))
def test_fuzzed_scenes_stay_bit_exact(seed):
    """Strict mode against the oracle on random scenes and cameras, counting and production kernel variants,
    head and checker shading."""
    R = _rt()
    rs = np.random.RandomState(1000 + seed)
    spheres, mats, camkw = _fuzz_scene(rs)
    flags = O.ACCUM_F32 | (O.SHADE_CHECKER if seed % 3 == 0 else 0)
    cam = scenes.camera_raw(48, 32, spp=3, frames=2, **camkw)
    for counters in (False, True):
        _strict_equals_oracle(R, 48, 32, 3, 2, flags, spheres, mats, cam, counters)


@pytest.mark.parametrize("flags", [0, O.ACCUM_F32, O.ACCUM_F32 | O.COMBINE_SUM, O.DISPATCH_FULL], ids=["rgba8", "f32", "f32_sum", "rgba8_full"])
def test_staged_batches_equal_pixel_sequential_batches_and_the_oracle(flags, monkeypatch):
    """A batch whose rank has few tiles per wave is rendered as (tile, frame) units into a staging buffer and
    folded into the image by a second kernel; RT_STAGED_FRAMES=0 keeps it pixel-sequential. Both must give
    the oracle's image and RNG states, also under a tile partition."""
    R = _rt()
    W, H, spp, frames = 104, 72, 4, 5
    spheres, mats = scenes.default_scene(1)
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames)
    for rank, world in ((0, 1), (1, 3)):
        img = O.OracleImage(W, H, flags)
        O.render(img, spheres, mats, cam, 0, frames, rank=rank, world=world)
        for staged in ("1", "0"):
            monkeypatch.setenv("RT_STAGED_FRAMES", staged)
            rt = _run_gpu(R, W, H, flags, spheres, mats, cam, frames, batched=True, rank=rank, world=world)
            assert np.array_equal(rt.read_rng_state(), img.rng), (staged, rank, world)
            if flags & O.ACCUM_F32:
                assert np.array_equal(rt.read_accum_f32().view(np.uint32), img.accum.view(np.uint32)), (staged, rank, world)
            assert np.array_equal(rt.read_rgba8(), img.view_rgba8()), (staged, rank, world)
            rt.close()


@pytest.mark.parametrize("group", ["2", "4", "8"])
@pytest.mark.parametrize("flags,scene", [(O.ACCUM_F32, "default"), (0, "default"), (O.ACCUM_F32 | O.SHADE_CHECKER, "bunch20"), (O.ACCUM_F32, "bunch300")],
                         ids=["f32_default", "rgba8_default", "f32_checker20", "f32_300"])
def test_frame_groups_of_staged_batches_change_no_bit(group, flags, scene, monkeypatch):
    """Staged launches hand a lane its pixel for a GROUP of consecutive frames (RT_FRAME_GROUP, a power of two; the default
    picks it from the launch shape and stays at 1 for images this small, hence the override). 7 frames: the last group is
    short for every group size; sky tiles (whole-wave branch) and traced tiles, the MASK, classic and LIST shapes, a partition."""
    R = _rt()
    W, H, spp, frames = 120, 72, 4, 7
    spheres, mats = scenes.default_scene(1) if scene == "default" else scenes.random_bunch(int(scene[5:]), 3, 3)
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames)
    monkeypatch.setenv("RT_FRAME_GROUP", group)
    for rank, world in ((0, 1), (2, 3)):
        img = O.OracleImage(W, H, flags)
        O.render(img, spheres, mats, cam, 0, frames, rank=rank, world=world)
        rt = _run_gpu(R, W, H, flags, spheres, mats, cam, frames, batched=True, rank=rank, world=world)
        assert np.array_equal(rt.read_rng_state(), img.rng), (rank, world)
        if flags & O.ACCUM_F32:
            assert np.array_equal(rt.read_accum_f32().view(np.uint32), img.accum.view(np.uint32)), (rank, world)
        assert np.array_equal(rt.read_rgba8(), img.view_rgba8()), (rank, world)
        rt.close()


def _render_state(R, W, H, flags, spheres, mats, cam, frames, batched, rank=0, world=1):
    rt = _run_gpu(R, W, H, flags, spheres, mats, cam, frames, batched=batched, rank=rank, world=world)
    out = (rt.read_rng_state().copy(), rt.read_accum_f32().view(np.uint32).copy() if flags & O.ACCUM_F32 else None,
           rt.read_rgba8().copy(), rt.sky_tile_stats())
    rt.close()
    return out


SKY_CASES = [
    # name, W, H, spp, frames, flags, scene, batched, (rank, world)
    ("default_rgba8", 640, 368, 2, 3, 0, "default", False, (0, 1)),
    ("default_f32_batch", 640, 368, 4, 3, O.ACCUM_F32, "default", True, (0, 1)),          # staged (tile, frame) units
    ("default_sum_partition", 512, 288, 2, 2, O.ACCUM_F32 | O.COMBINE_SUM, "default", False, (1, 3)),
    ("default_full_ragged", 500, 300, 1, 2, O.DISPATCH_FULL, "default", True, (0, 1)),     # 1 spp: pixel-sequential batch
    ("checker_bunch20", 768, 432, 2, 2, O.SHADE_CHECKER | O.ACCUM_F32, "bunch20", False, (0, 1)),
    ("bunch1024", 480, 272, 1, 2, 0, "bunch1024", False, (0, 1)),
]


@pytest.mark.parametrize("math", [0x0, 0x8], ids=["strict", "fast"])  # RT_MATH_STRICT, RT_MATH_FAST
@pytest.mark.parametrize("name,W,H,spp,frames,flags,scene,batched,part", SKY_CASES, ids=[c[0] for c in SKY_CASES])
def test_sky_tiles_change_nothing(name, W, H, spp, frames, flags, scene, batched, part, math, monkeypatch):
    """Tiles whose primary rays provably miss every sphere are finished without tracing (rt_sky_tiles_kernel).
    With RT_SKY_TILES=0 every tile is traced: image, float image and every RNG state must be identical, in both
    arithmetic modes, for per-frame, pixel-sequential and staged launches, and under a tile partition."""
    R = _rt()
    spheres, mats = _scene(scene)
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames)
    monkeypatch.setenv("RT_SKY_TILES", "1")
    on = _render_state(R, W, H, flags | math, spheres, mats, cam, frames, batched, *part)
    monkeypatch.setenv("RT_SKY_TILES", "0")
    off = _render_state(R, W, H, flags | math, spheres, mats, cam, frames, batched, *part)
    assert off[3] == (0, 0)
    sky, tiles = on[3]
    assert 0 < sky < tiles, (sky, tiles)
    assert np.array_equal(on[0], off[0])
    if flags & O.ACCUM_F32:
        assert np.array_equal(on[1], off[1])
    assert np.array_equal(on[2], off[2])


@pytest.mark.parametrize("seed", range(32))
def test_sky_tiles_fuzzed_cameras_and_scenes(seed, monkeypatch):
    """The sky-tile bound must be conservative for any scene and camera: random scenes (tiny and huge scales, cameras
    inside spheres, zero and negative radii) at a resolution where tiles are small enough to be sky."""
    R = _rt()
    rs = np.random.RandomState(5000 + seed)
    spheres, mats, camkw = _fuzz_scene(rs)
    W, H = 384, 224
    flags = O.ACCUM_F32 | (O.SHADE_CHECKER if seed % 3 == 0 else 0) | (0x8 if seed % 4 == 3 else 0)
    cam = scenes.camera_raw(W, H, spp=2, frames=2, **camkw)
    monkeypatch.setenv("RT_SKY_TILES", "1")
    on = _render_state(R, W, H, flags, spheres, mats, cam, 2, False)
    monkeypatch.setenv("RT_SKY_TILES", "0")
    off = _render_state(R, W, H, flags, spheres, mats, cam, 2, False)
    assert np.array_equal(on[0], off[0]) and np.array_equal(on[1], off[1]), on[3]


def test_sky_tiles_follow_the_camera_and_the_scene():
    """The flags are recomputed when the camera or the scene changes (a stale flag would skip a visible sphere)."""
    R = _rt()
    W, H = 512, 288
    spheres, mats = scenes.default_scene(1)
    rt = R.Raytracer(W, H, O.ACCUM_F32)
    rt.enable_rng_capture(True)
    tb = O.time_bits_table(1)
    seen = []
    for look, scene in (((0.0, 0.0, -1.0), (spheres, mats)), ((0.0, 0.6, -1.0), (spheres, mats)), ((0.0, 0.0, -1.0), (spheres[:3], mats))):
        rt.set_scene(*scene)
        cam = scenes.camera_raw(W, H, spp=1, frames=4, look=look)
        rt.clear(); rt.set_camera(cam); rt.set_time(int(tb[0])); rt.compute(); rt.sync()
        img = O.OracleImage(W, H, O.ACCUM_F32)
        O.render(img, scene[0], scene[1], cam, 0, 1)
        assert np.array_equal(rt.read_rng_state(), img.rng)
        assert np.array_equal(rt.read_accum_f32().view(np.uint32), img.accum.view(np.uint32))
        seen.append(rt.sky_tile_stats()[0])
    rt.close()
    assert len(set(seen)) == 3 and min(seen) > 0, seen


@pytest.mark.parametrize("look,expect", [((0.0, 0.0, -1.0), "mixed"), ((0.0, 1.0, 0.2), "all_sky"), ((0.0, -1.0, -0.05), "no_sky")])
def test_tile_order_hands_every_tile_out_once(look, expect):
    """The cost-ordered queue (more tiles than waves: 1024x768 = 12 288 tiles) sorts the traced tiles by their dearest pixel and spreads
    the sky tiles evenly between them - a permutation of the tiles whatever the split: a view with sky and spheres, one of nothing
    but sky, one of nothing but the ground. The first launch measures the cost, the later ones use the order; per-frame launches
    and a batched one, against the oracle."""
    R = _rt()
    W, H, frames = 1024, 768, 4
    spheres, mats = scenes.default_scene(1)
    cam = scenes.camera_raw(W, H, spp=1, frames=frames + 2, look=look)
    img = O.OracleImage(W, H, O.ACCUM_F32)
    cam_after, time_after = O.render(img, spheres, mats, cam, 0, frames)
    rt = _run_gpu(R, W, H, O.ACCUM_F32, spheres, mats, cam, frames)
    sky, tiles = rt.sky_tile_stats()
    assert tiles == 12288 and {"mixed": 0 < sky < tiles, "all_sky": sky == tiles, "no_sky": sky == 0}[expect], (sky, tiles)
    assert np.array_equal(rt.read_rng_state(), img.rng)
    assert np.array_equal(rt.read_accum_f32().view(np.uint32), img.accum.view(np.uint32))
    # two more frames as one batched launch on the same context (the order is in place)
    cam2 = np.array(cam, copy=True); cam2["current_frame"] = frames
    rt.set_camera(cam2)
    rt.render_frames(O.time_bits_table(frames + 2)[frames:])
    rt.sync()
    O.render(img, spheres, mats, cam_after, time_after, 2)
    assert np.array_equal(rt.read_rng_state(), img.rng)
    assert np.array_equal(rt.read_accum_f32().view(np.uint32), img.accum.view(np.uint32))
    rt.close()


@pytest.mark.parametrize("flags", [0, O.ACCUM_F32, O.ACCUM_F32 | O.COMBINE_SUM], ids=["rgba8", "f32", "f32_sum"])
def test_deferred_passes_equal_single_launches_and_the_oracle(flags, monkeypatch):
    """rt_compute records a pass; up to RT_DEFER_FRAMES (default 64) recorded passes run as one batched launch when
    the caller synchronises or changes what they depend on. 11 passes (8 + 3), non-consecutive frame indices, a new
    spp after pass 4 and a new scene after pass 7 must give the oracle's image and RNG states, as RT_DEFER_FRAMES=1 does."""
    R = _rt()
    W, H, frames = 200, 120, 11
    scene_a, scene_b = scenes.default_scene(1), scenes.random_bunch(6, 3, 9)
    tb = O.time_bits_table(frames)
    frame_ix = [0, 1, 2, 3, 5, 6, 7, 9, 10, 11, 12]  # a frame-set split skips indices

    def plan(f):
        return (2 if f < 4 else 3), (scene_a if f < 7 else scene_b)

    img = O.OracleImage(W, H, flags)
    for f in range(frames):
        spp, (sp, mt) = plan(f)
        cam = scenes.camera_raw(W, H, spp=spp, frames=64, current_frame=frame_ix[f])
        O.compute(img, sp, mt, cam, int(tb[f]))
    results = []
    for defer in ("64", "24", "8", "1"):
        monkeypatch.setenv("RT_DEFER_FRAMES", defer)
        rt = R.Raytracer(W, H, flags)
        rt.enable_rng_capture(True)
        last_scene = None
        for f in range(frames):
            spp, sc = plan(f)
            if sc is not last_scene:
                rt.set_scene(*sc)
                last_scene = sc
            rt.set_time(int(tb[f]))
            rt.set_camera(scenes.camera_raw(W, H, spp=spp, frames=64, current_frame=frame_ix[f]))
            rt.compute()
        rt.sync()
        results.append((rt.read_rng_state().copy(), rt.read_rgba8().copy(),
                        rt.read_accum_f32().view(np.uint32).copy() if flags & O.ACCUM_F32 else None))
        rt.close()
    for rng, rgba, acc in results:
        assert np.array_equal(rng, img.rng)
        assert np.array_equal(rgba, img.view_rgba8())
        if flags & O.ACCUM_F32:
            assert np.array_equal(acc, img.accum.view(np.uint32))


def test_deferred_passes_are_flushed_by_observers():
    """clear(), a partition change and reads between recorded passes act on the passes recorded before them."""
    R = _rt()
    W, H = 96, 64
    sp, mt = scenes.default_scene(1)
    tb = O.time_bits_table(4)
    rt = R.Raytracer(W, H, O.ACCUM_F32)
    rt.set_scene(sp, mt)
    img = O.OracleImage(W, H, O.ACCUM_F32)
    def one(f, rank, world):
        cam = scenes.camera_raw(W, H, spp=2, frames=8, current_frame=f)
        rt.set_time(int(tb[f])); rt.set_camera(cam); rt.compute()
        O.compute(img, sp, mt, cam, int(tb[f]), rank=rank, world=world)
    one(0, 0, 1)
    rt.clear()                      # wipes the image AFTER pass 0 ran
    img.accum[:] = 0
    one(1, 0, 1)
    rt.set_partition(1, 2)          # pass 1 covered every tile, pass 2 and 3 only rank 1's
    one(2, 1, 2)
    one(3, 1, 2)
    assert np.array_equal(rt.read_accum_f32().view(np.uint32), img.accum.view(np.uint32))
    rt.close()


def _bound_image_setup(R, frames):
    import torch

    from raytracer_amd import dist as rdist

    W, H, spp, flags = 96, 64, 2, O.ACCUM_F32
    sp, mt = scenes.default_scene(1)
    rt = R.Raytracer(W, H, flags)
    rt.set_scene(sp, mt)
    img = rdist.DeviceImage(rt, "cuda:0")
    ref = O.OracleImage(W, H, flags)
    O.render(ref, sp, mt, scenes.camera_raw(W, H, spp=spp, frames=frames), 0, frames)
    tb = O.time_bits_table(frames)

    def record(f):
        rt.set_time(int(tb[f])); rt.set_camera(scenes.camera_raw(W, H, spp=spp, frames=frames, current_frame=f)); rt.compute()

    return torch, rt, img, ref, record


@pytest.mark.parametrize("how", ["accum_device_ptr", "flush"])
def test_recorded_passes_reach_a_bound_image_without_an_observing_read(how):
    """rt_compute only records (deferred passes). A caller that owns the image must see the finished picture after
    rt_accum_device_ptr or rt_flush plus a device synchronisation of its own - no rt_sync, no rt_read_*."""
    R = _rt()
    torch, rt, img, ref, record = _bound_image_setup(R, 3)
    for f in range(3):
        record(f)
    torch.cuda.synchronize()
    assert not img.tensor.any().item(), "nothing has been launched yet: the passes are only recorded"
    if how == "accum_device_ptr":
        ptr, nbytes = rt.accum_device_ptr()  # launches the recorded passes
        assert ptr == img.tensor.data_ptr() and nbytes == img.tensor.numel() * 4
    else:
        rt.flush()
    torch.cuda.synchronize()
    assert np.array_equal(img.tensor.cpu().numpy().view(np.uint32), ref.accum.view(np.uint32))
    rt.bind_accum(0, 0)
    rt.close()


def test_passes_on_a_callers_stream_are_enqueued_by_rt_compute_itself():
    """rt_set_stream: no deferral - work the caller enqueues on its stream after rt_compute runs after the pass."""
    R = _rt()
    torch, rt, img, ref, record = _bound_image_setup(R, 3)
    stream = torch.cuda.Stream()
    rt.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        for f in range(3):
            record(f)
        copy = img.tensor.clone()  # enqueued on the same stream, after the three passes
    stream.synchronize()
    assert np.array_equal(copy.cpu().numpy().view(np.uint32), ref.accum.view(np.uint32))
    rt.set_stream(0)
    rt.bind_accum(0, 0)
    rt.close()


def test_headless_cli_per_frame_equals_batched(tmp_path):
    """rt_headless --per-frame (App::update / App::render per redraw) renders what the batched run renders - also for
    spp != 1, which needs the camera upload after --spp / --frames are applied (frame 0 used the constructor's)."""
    import os, subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "rt_headless")
    base = [exe, "--width", "96", "--height", "64", "--spp", "2", "--frames", "3", "--scene", "default", "--seed", "1"]
    a, b = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    subprocess.run(base + ["--out", a], check=True, timeout=120, capture_output=True)
    subprocess.run(base + ["--per-frame", "--out", b], check=True, timeout=120, capture_output=True)
    assert open(a, "rb").read() == open(b, "rb").read()


def test_headless_png_with_presentation_equals_the_oracle_through_the_screen_pass(tmp_path):
    """rt_headless --present --out x.png on the GPU, decoded, against the ORACLE's RGBA8 image pushed
    through the reference's screen pass on the CPU: pow(c, 2.2) (screen_shader.wgsl:36-39), then the sRGB encode of
    the Bgra8UnormSrgb surface (render_env.rs:48). The plain PNG must equal the oracle's bytes exactly; the presented
    one within 1/255 (two roundings: ours and the formula's)."""
    import os, subprocess

    from PIL import Image

    from raytracer_amd import host

    W, H, spp, frames = 128, 64, 2, 4
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "rt_headless")
    base = [exe, "--width", str(W), "--height", str(H), "--spp", str(spp), "--frames", str(frames), "--scene", "default", "--seed", "1"]
    plain, shown = str(tmp_path / "plain.png"), str(tmp_path / "shown.png")
    subprocess.run(base + ["--out", plain], check=True, timeout=120, capture_output=True)
    subprocess.run(base + ["--present", "--out", shown], check=True, timeout=120, capture_output=True)
    sp, mt = host.scene_default(1)  # the scene the CLI builds (same seeded generator)
    ref = O.OracleImage(W, H, 0)
    O.render(ref, sp, mt, host.camera_raw(W, H, spp=spp, frames=frames), 0, frames)
    assert ref.rgba8[..., :3].any()
    assert np.array_equal(np.asarray(Image.open(plain).convert("RGBA")), ref.rgba8)
    lin = (ref.rgba8[..., :3] / 255.0) ** 2.2
    srgb = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * lin ** (1 / 2.4) - 0.055)
    got = np.asarray(Image.open(shown).convert("RGB")).astype(int)
    assert np.abs(got - np.rint(srgb * 255)).max() <= 1


def test_moved_zoomed_host_camera_through_the_app_equals_the_oracle():
    """the host mirror's Camera driven like the reference's (W / D / Space keys, zoom, look-at-pixel,
    K for 2 spp: camera.rs:133-206), rendered by the host App on the GPU, bit for bit against the oracle fed THE
    SAME 160 bytes - also after a second move mid-render (scene_was_updated restarts at frame 0 but keeps the image:
    the reference never clears color_cache, and cw = 0 at frame 0 overwrites it)."""
    R = _rt()
    from raytracer_amd import host

    W, H, frames = 160, 96, 3
    sp, mt = host.scene_default(1)
    app = host.App(W, H, R.RT_ACCUM_RGBA8, sp, mt, spp=1, frames=frames)
    app.enable_rng_capture(True)
    for k in ("W", "W", "D", "Space", "K"):  # forward twice, right, up; spp 1 -> 2
        app.key(k)
    app.zoom(1.5)
    app.look_at_pixel(37.0, -11.0)
    cam = app.camera_raw()
    default = host.camera_raw(W, H, spp=2, frames=frames)
    assert int(cam["samples_per_pixel"]) == 2 and int(cam["current_frame"]) == 0
    assert not np.array_equal(cam["pixel_to_world"], default["pixel_to_world"]) and not np.array_equal(cam["pos"], default["pos"])
    ref = O.OracleImage(W, H, 0)
    app.redraw(frames)  # per-frame launches through App::update / App::render
    _, tbits = O.render(ref, sp, mt, cam, 0, frames)
    assert np.array_equal(app.read_rgba8(), ref.rgba8) and np.array_equal(app.read_rng_state(), ref.rng)
    # move again mid-life: frame counter back to 0, time keeps running (time.rs is never reset)
    app.key("A"); app.zoom(-0.5)
    cam2 = app.camera_raw()
    assert int(cam2["current_frame"]) == 0 and app.time_bits == tbits
    app.run(frames)  # batched this time
    O.render(ref, sp, mt, cam2, tbits, frames)
    assert np.array_equal(app.read_rgba8(), ref.rgba8) and np.array_equal(app.read_rng_state(), ref.rng)
    app.close()


@pytest.mark.parametrize("n,W,H,fov", [(100, 256, 160, 0.25), (300, 320, 192, 0.1), (1024, 192, 128, 0.02)],
                         ids=["100_wide", "300_mid", "1024_narrow"])
def test_list_shape_candidate_lists_short_long_and_overflowing(n, W, H, fov, monkeypatch):
    """Scenes of more than 64 spheres trace their primary rays against per-tile candidate LISTS (<= 15 entries, else
    the general trace): a wide view makes most lists overflow, a narrow one makes them short. Bit-exact against the
    oracle either way, and equal to the kernel without the split."""
    R = _rt()
    sp, mt = scenes.random_bunch(n, 5, 11)
    frames = 3
    cam = scenes.camera_raw(W, H, spp=3, frames=frames, fov=fov)
    img = O.OracleImage(W, H, O.ACCUM_F32)
    O.render(img, sp, mt, cam, 0, frames)
    for split in ("1", "0"):
        monkeypatch.setenv("RT_SPLIT", split)
        for batched in (False, True):
            rt = _run_gpu(R, W, H, O.ACCUM_F32, sp, mt, cam, frames, batched=batched)
            assert np.array_equal(rt.read_rng_state(), img.rng), (split, batched)
            assert np.array_equal(rt.read_accum_f32().view(np.uint32), img.accum.view(np.uint32)), (split, batched)
            rt.close()


@pytest.mark.parametrize("case", ["row_of_100", "two_clumps_of_40", "huge_shell_around_200"])
def test_candidate_queue_of_the_general_trace_fills_and_overflows(case, monkeypatch):
    """Strict launches on scenes of more than 64 spheres queue a ray's candidates (four per lane) across the 32-sphere
    blocks and test them afterwards. Rays that pierce many spheres - a row of spheres behind each other, clumps that
    put more than four candidates into one block, a camera inside a huge sphere (every ray hits it, from inside) -
    fill the queue, drain it early and fall back to the block-by-block walk: the image stays the oracle's, with
    tile lists and without (RT_SPLIT=0: primary rays take the general trace too)."""
    R = _rt()
    rs = np.random.RandomState(91)
    mats = scenes.make_materials(rs.random_sample((3, 4)).astype(np.float32))
    if case == "row_of_100":
        centers = [(0.05 * (k % 3), 0.03 * (k % 5), -3.0 - 0.45 * k) for k in range(100)]
        radii = [1.0 + 0.002 * k for k in range(100)]
    elif case == "two_clumps_of_40":
        centers = [tuple(rs.normal(0, 0.15, 3) + np.array([-0.6 if k < 40 else 0.6, 0.0, -4.0])) for k in range(80)]
        radii = [0.5] * 80
    else:
        centers = [(0.0, 0.0, 0.0)] + [tuple((rs.random_sample(3) * 2 - 1) * 6 + np.array([0, 0, -9.0])) for _ in range(199)]
        radii = [40.0] + [0.7] * 199
    n = len(radii)
    spheres = scenes.make_spheres(centers, radii, rs.randint(0, 3, size=n))
    W, H = 192, 128
    cam = scenes.camera_raw(W, H, spp=2, frames=2)
    _strict_equals_oracle(R, W, H, 2, 2, O.ACCUM_F32, spheres, mats, cam, False)
    monkeypatch.setenv("RT_SPLIT", "0")
    _strict_equals_oracle(R, W, H, 2, 2, O.ACCUM_F32, spheres, mats, cam, False)


def test_sky_tiles_are_off_for_a_non_finite_camera(monkeypatch):
    """A non-finite pixel_to_world entry - even in the column the shader multiplies by 0.0 - makes every ray NaN, and
    NaN roots pass hit_sphere's range tests (wgsl:214-222): such rays hit. No tile may then be treated as sky."""
    R = _rt()
    W, H = 256, 160
    sp, mt = scenes.default_scene(1)
    cam = scenes.camera_raw(W, H, spp=1, frames=2, look=(0.0, 0.6, -1.0))
    cam["pixel_to_world"][8] = np.inf
    monkeypatch.setenv("RT_SKY_TILES", "1")
    on = _render_state(R, W, H, O.ACCUM_F32, sp, mt, cam, 1, False)
    assert on[3][0] == 0 and on[3][1] > 0
    monkeypatch.setenv("RT_SKY_TILES", "0")
    off = _render_state(R, W, H, O.ACCUM_F32, sp, mt, cam, 1, False)
    assert np.array_equal(on[0], off[0]) and np.array_equal(on[1], off[1])


@pytest.mark.parametrize("knob,value", [("RT_REJECT_TRIES", "1"), ("RT_REJECT_TRIES", "5"), ("RT_BATCH_LANES", "1"), ("RT_BATCH_LANES", "9"),
                                        ("RT_WAVES_PER_CU", "3"), ("RT_WAVES_PER_CU", "32"), ("RT_STAGED_FRAMES", "0"),
                                        ("RT_TILE_ORDER", "0"), ("RT_DEFER_FRAMES", "3"), ("RT_SPLIT", "0"), ("RT_SKY_TILES", "0"), ("RT_FRAME_GROUP", "4"),
                                        ("RT_QUEUE_CHUNK", "5"), ("RT_QUEUE_CHUNK", "1000"), ("RT_STAGE_MIN_SPP", "1"), ("RT_QUEUE_STATIC", "64"), ("RT_QUEUE_STATIC", "0")])
def test_launch_knobs_change_no_bit(knob, value, monkeypatch):
    """Launch width, rejection rounds per iteration, parked lanes per service, staging, tile order, the size of a
    deferred batch and how units reach the waves (several per atomic on the queue head, or strided without a queue) only move work between lanes and launches: each pixel consumes its own RNG stream in the
    reference's order, so image and RNG states stay the oracle's."""
    R = _rt()
    W, H, spp, frames = 160, 96, 4, 5
    sp, mt = scenes.random_bunch(20, 3, 3)
    flags = O.ACCUM_F32 | O.SHADE_CHECKER
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames)
    img = O.OracleImage(W, H, flags)
    O.render(img, sp, mt, cam, 0, frames)
    monkeypatch.setenv(knob, value)
    rt = _run_gpu(R, W, H, flags, sp, mt, cam, frames)
    assert np.array_equal(rt.read_rng_state(), img.rng)
    assert np.array_equal(rt.read_accum_f32().view(np.uint32), img.accum.view(np.uint32))
    rt.close()


def _grid_scene(case):
    rs = np.random.RandomState(131)
    mats = scenes.make_materials(rs.random_sample((3, 4)).astype(np.float32))
    camkw, p2w_zero_row, expect_grid = {}, None, True
    if case == "camera_inside_the_bunch":          # origins inside the grid's box: the walk starts in the middle of it
        spheres, mats = scenes.random_bunch(400, 3, 21)
        camkw = dict(pos=(1.5, -2.0, -30.0), look=(0.3, 0.1, -1.0), fov=0.4)
    elif case == "camera_beyond_the_grids_reach":  # |o|^2 > grid_o2max: primary rays loop over all spheres, bounce rays walk the grid
        spheres, mats = scenes.random_bunch(200, 3, 22)
        camkw = dict(pos=(0.0, 0.0, 400.0), fov=0.03)
    elif case == "every_sphere_twice":             # equal roots out of index order: the lower index must win (mirror vs diffuse differ)
        s, mats = scenes.random_bunch(75, 3, 23)
        spheres = np.concatenate([s, s])           # sphere i and i + 75: different index parity
    elif case == "flat_sheet":                     # one layer of cells along z
        centers = [((k % 15) * 0.9 - 6.3, (k // 15) * 0.9 - 4.5, -14.0) for k in range(150)]
        spheres = scenes.make_spheres(centers, [0.5] * 150, rs.randint(0, 3, size=150))
    elif case == "rays_parallel_to_an_axis":       # a camera whose rays all have dx == 0: the slab of that axis has no crossings
        spheres, mats = scenes.random_bunch(120, 3, 24)
        p2w_zero_row = 0
    elif case == "mixed_radii_and_signs":          # negative radii are listed by |r|; tiny and big spheres share cells
        centers = (rs.random_sample((260, 3)) * 2 - 1) * np.array([9.0, 6.0, 6.0]) + np.array([0.0, 0.0, -16.0])
        radii = rs.choice([0.05, 0.3, -0.4, 0.8, 2.5], size=260)
        spheres = scenes.make_spheres(centers, radii, rs.randint(0, 3, size=260))
    elif case == "big_spheres_everywhere_no_grid":  # every sphere touches most cells: rt_set_scene declines the grid
        centers = (rs.random_sample((90, 3)) * 2 - 1) * 2.0 + np.array([0.0, 0.0, -30.0])
        spheres = scenes.make_spheres(centers, [6.0] * 90, rs.randint(0, 3, size=90))
        expect_grid = False
    else:
        raise KeyError(case)
    return spheres, mats, camkw, p2w_zero_row, expect_grid


@pytest.mark.parametrize("case", ["camera_inside_the_bunch", "camera_beyond_the_grids_reach", "every_sphere_twice", "flat_sheet",
                                  "rays_parallel_to_an_axis", "mixed_radii_and_signs", "big_spheres_everywhere_no_grid"])
def test_grid_trace_of_big_scenes_changes_no_bit(case, monkeypatch, capfd):
    """Scenes of more than 64 spheres: each lane walks the cells of a uniform grid along its ray and filters only the
    spheres listed there (rt_api.cpp build_grid, the GRID TRACE of rt_kernels.hip). Spheres arrive out of index order and
    possibly twice, so hit_sphere's tie rule (wgsl:214-222: the lowest index among equal roots) is applied by index. The
    strict image is the oracle's with tile lists and without (RT_SPLIT=0: primary rays walk the grid too); in fast
    arithmetic grid on / off give the same bits (the filter is the same, only the set of spheres it runs on shrinks)."""
    R = _rt()
    spheres, mats, camkw, zero_row, expect_grid = _grid_scene(case)
    W, H, spp, frames = 192, 128, 2, 2
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames, **camkw)
    if zero_row is not None:
        cam["pixel_to_world"][zero_row::4] = 0.0   # row `zero_row` of the column-major matrix: that component of every direction is +-0
    monkeypatch.setenv("RT_GRID_VERBOSE", "1")
    grid_on = "2" if expect_grid else "1"  # 2: build the grid although the scene is below the size from which it pays (rt_api.cpp, build_grid)
    monkeypatch.setenv("RT_GRID", grid_on)
    capfd.readouterr()
    for split in ("1", "0"):
        monkeypatch.setenv("RT_SPLIT", split)
        _strict_equals_oracle(R, W, H, spp, frames, O.ACCUM_F32, spheres, mats, cam, False)
    said = capfd.readouterr().err
    assert ("rt_set_scene: grid " in said) == expect_grid and ("rt_set_scene: no grid" in said) == (not expect_grid), said
    monkeypatch.delenv("RT_SPLIT")
    for math in (0x0, 0x8):
        monkeypatch.setenv("RT_GRID", grid_on)
        on = _render_state(R, W, H, O.ACCUM_F32 | math, spheres, mats, cam, frames, True)
        monkeypatch.setenv("RT_GRID", "0")
        off = _render_state(R, W, H, O.ACCUM_F32 | math, spheres, mats, cam, frames, True)
        assert np.array_equal(on[0], off[0]) and np.array_equal(on[1], off[1]), math


def test_grid_trace_of_a_scene_with_more_than_65535_spheres():
    """70 000 spheres: no tile lists (their entries are 16 bits), so every ray - primary rays too - takes the general trace, walks
    the grid and tests its candidates where it meets them (the candidate queue's entries are 16 bits as well). Bit-exact
    against the oracle; without the grid this scene costs 70 000 filter tests per ray."""
    R = _rt()
    spheres, mats = scenes.random_bunch(70000, 3, 41)
    W, H, spp, frames = 64, 48, 1, 2
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames, pos=(0.0, 0.0, -300.0), fov=0.3)
    _strict_equals_oracle(R, W, H, spp, frames, O.ACCUM_F32, spheres, mats, cam, False)
