"""rt_render_frames_static — the entry point that allocates nothing (include/rt_hip.h) — against rt_render_frames.

Called through the C ABI on two inputs in a row, the second smaller than the first (fewer tiles, fewer frames, fewer
spheres), so that the second call runs over the first call's leftovers in the per-GPU static scratch. Bar: bit-exact
(RGBA8 / float image and every pixel's final RNG state), as for every strict-mode result. Collected last on purpose:
the round-end run uses `pytest -x`.
"""
import numpy as np
import pytest

import oracle_lib as O
import scenes

pytestmark = pytest.mark.gpu


def _render(R, static, W, H, flags, spheres, mats, spp, frames, calls, partition=None):
    """`calls` = frame counts of consecutive render calls on one context. Returns (rgba8, f32 or None, rng)."""
    rt = R.Raytracer(W, H, flags)
    if partition:
        rt.set_partition(*partition)
    rt.set_scene(spheres, mats)
    rt.enable_rng_capture(True)
    rt.set_camera(scenes.camera_raw(W, H, spp=spp, frames=frames))
    tb = O.time_bits_table(frames)
    at = 0
    for n in calls:
        (rt.render_frames_static if static else rt.render_frames)(tb[at:at + n])
        at += n
    rt.sync()
    out = (rt.read_rgba8(), rt.read_accum_f32() if flags & R.RT_ACCUM_F32 else None, rt.read_rng_state())
    rt.close()
    return out


def _same(a, b):
    assert np.array_equal(a[0], b[0]), "RGBA8 image differs"
    if a[1] is not None:
        assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), "float image differs"
    assert np.array_equal(a[2], b[2]), "final RNG states differ"


def test_static_entry_point_matches_render_frames_twice_in_a_row_second_call_smaller():
    import raytracer_amd as R

    big = dict(W=272, H=208, flags=R.RT_ACCUM_F32 | R.RT_SHADE_CHECKER, spp=4, frames=5)
    small = dict(W=96, H=64, flags=R.RT_ACCUM_RGBA8, spp=2, frames=3)
    s_big, s_small = scenes.random_bunch(20, 3, 3), scenes.default_scene(1)
    # first call: 884 tiles, 5 frames, 20 spheres; second: 96 tiles, 3 frames, 4 spheres - over the first call's leftovers
    got_big = _render(R, True, spheres=s_big[0], mats=s_big[1], calls=[5], **big)
    got_small = _render(R, True, spheres=s_small[0], mats=s_small[1], calls=[3], **small)
    _same(got_big, _render(R, False, spheres=s_big[0], mats=s_big[1], calls=[5], **big))
    _same(got_small, _render(R, False, spheres=s_small[0], mats=s_small[1], calls=[3], **small))
    # and against the oracle itself
    img = O.OracleImage(small["W"], small["H"], 0)
    O.render(img, s_small[0], s_small[1], scenes.camera_raw(small["W"], small["H"], spp=small["spp"], frames=small["frames"]), 0, small["frames"])
    assert np.array_equal(got_small[0], img.rgba8) and np.array_equal(got_small[2], img.rng)


def test_static_entry_point_split_calls_on_one_context_and_past_the_end():
    import raytracer_amd as R

    sp, mt = scenes.default_scene(1)
    kw = dict(W=80, H=48, flags=R.RT_ACCUM_RGBA8, spheres=sp, mats=mt, spp=3, frames=4)
    # 2 + 1 + 3 frames on one context: single-frame launches take the kernarg path, the last call runs past frames_to_render
    _same(_render(R, True, calls=[2, 1, 3], **{**kw, "frames": 4}), _render(R, False, calls=[2, 1, 3], **{**kw, "frames": 4}))


def test_static_entry_point_rejects_what_it_cannot_hold():
    import raytracer_amd as R

    sp, mt = scenes.default_scene(1)
    rt = R.Raytracer(64, 48, R.RT_ACCUM_RGBA8)
    rt.set_scene(sp, mt)
    rt.set_camera(scenes.camera_raw(64, 48, spp=1, frames=1000))
    with pytest.raises(R.RtError) as e:
        rt.render_frames_static(O.time_bits_table(257))
    assert e.value.code == R._abi.RT_ERR_INVALID and "256" in str(e.value)
    rt.set_time(int(O.time_bits_table(1)[0]))
    rt.compute()  # recorded, not launched
    with pytest.raises(R.RtError) as e:
        rt.render_frames_static(O.time_bits_table(2))
    assert e.value.code == R._abi.RT_ERR_STATE and "rt_sync" in str(e.value)
    rt.sync()
    rt.render_frames_static(O.time_bits_table(2))  # fine now
    rt.sync()
    rt.close()


def test_static_entry_point_allocates_no_device_memory_big_scene_and_partition():
    """check the property the entry point exists for. After one warm-up call (the first launch of a kernel loads
    its code object), free device memory must not move over further static calls - on the MASK path (20 spheres), on the
    classic path of a scene above 64 spheres (all-ones masks from the scratch, no lists; at 400 spheres its general trace walks
    the grid rt_set_scene built) and under a tile partition."""
    import torch

    import raytracer_amd as R

    cases = [dict(W=272, H=208, flags=R.RT_ACCUM_F32 | R.RT_SHADE_CHECKER, spp=2, frames=64, scene=scenes.random_bunch(20, 3, 3), partition=None),
             dict(W=160, H=128, flags=R.RT_ACCUM_F32, spp=2, frames=64, scene=scenes.random_bunch(200, 4, 16), partition=None),
             dict(W=160, H=128, flags=R.RT_ACCUM_F32, spp=2, frames=64, scene=scenes.random_bunch(400, 4, 17), partition=None),  # big enough for the grid trace
             dict(W=272, H=208, flags=R.RT_ACCUM_RGBA8, spp=2, frames=64, scene=scenes.default_scene(1), partition=(1, 3))]
    for c in cases:
        sp, mt = c["scene"]
        rt = R.Raytracer(c["W"], c["H"], c["flags"])
        if c["partition"]:
            rt.set_partition(*c["partition"])
        rt.set_scene(sp, mt)
        rt.enable_rng_capture(True)
        rt.set_camera(scenes.camera_raw(c["W"], c["H"], spp=c["spp"], frames=c["frames"]))
        tb = O.time_bits_table(c["frames"])
        rt.render_frames_static(tb[:3])  # warm-up
        rt.sync()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        at = 3
        for n in (1, 7, 2, 5):
            rt.render_frames_static(tb[at:at + n])
            at += n
        rt.sync()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0, "a static call changed the free device memory"
        got = (rt.read_rgba8(), rt.read_accum_f32() if c["flags"] & R.RT_ACCUM_F32 else None, rt.read_rng_state())
        rt.close()
        _same(got, _render(R, False, c["W"], c["H"], c["flags"], sp, mt, c["spp"], c["frames"], [at], partition=c["partition"]))


def test_two_contexts_alternate_static_calls_without_syncing():
    """every context of a GPU shares that GPU's scratch and has a stream of its own; the library orders the
    calls on the device (one event per GPU), so alternating calls of two contexts need no rt_sync in between."""
    import raytracer_amd as R

    a = dict(W=272, H=208, flags=R.RT_ACCUM_F32 | R.RT_SHADE_CHECKER, spp=3, frames=12, scene=scenes.random_bunch(20, 3, 3))
    b = dict(W=208, H=144, flags=R.RT_ACCUM_RGBA8, spp=2, frames=12, scene=scenes.default_scene(1))
    ctxs = []
    for c in (a, b):
        rt = R.Raytracer(c["W"], c["H"], c["flags"])
        rt.set_scene(*c["scene"])
        rt.enable_rng_capture(True)
        rt.set_camera(scenes.camera_raw(c["W"], c["H"], spp=c["spp"], frames=c["frames"]))
        ctxs.append(rt)
    tb = O.time_bits_table(12)
    for at in range(0, 12, 3):  # A, B, A, B, ... : four calls each, none waited for
        for rt in ctxs:
            rt.render_frames_static(tb[at:at + 3])
    for c, rt in zip((a, b), ctxs):
        rt.sync()
        got = (rt.read_rgba8(), rt.read_accum_f32() if c["flags"] & R.RT_ACCUM_F32 else None, rt.read_rng_state())
        rt.close()
        _same(got, _render(R, False, c["W"], c["H"], c["flags"], c["scene"][0], c["scene"][1], c["spp"], c["frames"], [12]))


def test_static_entry_point_under_graph_capture_replays_with_its_own_frame_table():
    """the header promises that rt_render_frames_static can be captured. Capture a 4-frame call on a caller's
    stream (nothing runs), replay it, compare with rt_render_frames; then clear the image, make ANOTHER static call on a
    second context (which rewrites the scratch's frame table) and replay again: the graph carries its own table."""
    import torch

    import raytracer_amd as R

    W, H, spp = 272, 208, 3
    sp, mt = scenes.random_bunch(20, 3, 3)
    flags = R.RT_ACCUM_F32 | R.RT_SHADE_CHECKER
    tb = O.time_bits_table(16)
    want = _render(R, False, W, H, flags, sp, mt, spp, 16, [4])

    rt = R.Raytracer(W, H, flags)
    rt.set_scene(sp, mt)
    rt.enable_rng_capture(True)
    cam = scenes.camera_raw(W, H, spp=spp, frames=16)
    rt.set_camera(cam)
    rt.render_frames_static(tb[:4])  # warm-up outside the capture (code objects load on first launch)
    rt.sync()
    rt.clear()
    rt.sync()
    cam["current_frame"] = 0
    rt.set_camera(cam)
    stream = torch.cuda.Stream()
    rt.set_stream(stream.cuda_stream)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        rt.render_frames_static(tb[:4])
    torch.cuda.synchronize()
    assert not rt.read_accum_f32().any(), "capturing must not run anything"
    graph.replay()
    torch.cuda.synchronize()
    _same((rt.read_rgba8(), rt.read_accum_f32(), rt.read_rng_state()), want)

    other = R.Raytracer(96, 64, R.RT_ACCUM_RGBA8)  # another user of this GPU's scratch: different frames, sizes, table
    other.set_scene(*scenes.default_scene(1))
    other.set_camera(scenes.camera_raw(96, 64, spp=1, frames=16))
    other.render_frames_static(tb[8:16])
    other.sync()
    other.close()
    rt.clear()
    rt.sync()
    graph.replay()
    torch.cuda.synchronize()
    _same((rt.read_rgba8(), rt.read_accum_f32(), rt.read_rng_state()), want)
    rt.set_stream(0)
    rt.close()
