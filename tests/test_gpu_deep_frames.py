"""Deep renders: BASELINE config 5's shape (1 spp, 1024 frames, float sum, frame sets) at a size the oracle renders in
seconds. Pins what the 16-frame C5 digest cannot: time bits by 1024 repeated f32 additions of 0.01 (time.rs:72-77; bits
up to 0x4123....), cw = f32(frame) / f32(frame + 1) within 2^-10 of 1 (wgsl:108-109), and batches of 64 recorded passes.
Bar: bit-exact float image and final RNG state of every pixel (RT_MATH_STRICT)."""
import numpy as np
import pytest

import oracle_lib as O
import scenes

pytestmark = pytest.mark.gpu

W, H, FRAMES = 320, 192, 1024


def _oracle(flags, frames, cam, tb, spheres, mats):
    img = O.OracleImage(W, H, flags)
    for f in frames:
        c = np.array(cam, copy=True)
        c["current_frame"] = f
        O.compute(img, spheres, mats, c, int(tb[f]))
    return img


def test_1024_frames_float_sum_in_batches_of_64():
    import raytracer_amd as R

    sp, mt = scenes.default_scene(1)
    cam = scenes.camera_raw(W, H, spp=1, frames=FRAMES)
    tb = O.time_bits_table(FRAMES)
    assert tb[-1] >> 24 == 0x41  # the time has passed 8.0: mantissa steps of 0.01 no longer land where they did near 0
    rt = R.Raytracer(W, H, R.RT_ACCUM_F32 | R.RT_COMBINE_SUM)
    rt.set_scene(sp, mt)
    rt.enable_rng_capture(True)
    rt.set_camera(cam)
    for at in range(0, FRAMES, 64):
        rt.render_frames(tb[at:at + 64])
    rt.sync()
    got, rng = rt.read_accum_f32(), rt.read_rng_state()
    rt.close()
    want = _oracle(O.ACCUM_F32 | O.COMBINE_SUM, range(FRAMES), cam, tb, sp, mt)
    assert np.array_equal(got.view(np.uint32), want.accum.view(np.uint32))
    assert np.array_equal(rng, want.rng)
    assert np.all(got[: H // 16 * 16, : W // 16 * 16, 3] == FRAMES)


def test_frame_set_3_mod_8_of_1024_recorded_pass_by_pass():
    """rank 3 of an 8-GPU frame split: every 8th frame, recorded by rt_compute, batched by the library"""
    import raytracer_amd as R

    sp, mt = scenes.default_scene(1)
    cam = scenes.camera_raw(W, H, spp=1, frames=FRAMES)
    tb = O.time_bits_table(FRAMES)
    mine = list(range(3, FRAMES, 8))
    rt = R.Raytracer(W, H, R.RT_ACCUM_F32 | R.RT_COMBINE_SUM)
    rt.set_scene(sp, mt)
    rt.enable_rng_capture(True)
    for f in mine:
        cam["current_frame"] = f
        rt.set_camera(cam)
        rt.set_time(int(tb[f]))
        rt.compute()
    rt.sync()
    got, rng = rt.read_accum_f32(), rt.read_rng_state()
    rt.close()
    want = _oracle(O.ACCUM_F32 | O.COMBINE_SUM, mine, cam, tb, sp, mt)
    assert np.array_equal(got.view(np.uint32), want.accum.view(np.uint32))
    assert np.array_equal(rng, want.rng)


@pytest.mark.parametrize("accum", ["f32", "rgba8"])
def test_1024_frames_running_average(accum):
    """the reference's own recurrence (wgsl:106-112) that deep: cw within 2^-10 of 1; RGBA8 re-quantised every frame"""
    import raytracer_amd as R

    sp, mt = scenes.default_scene(1)
    cam = scenes.camera_raw(W, H, spp=1, frames=FRAMES)
    tb = O.time_bits_table(FRAMES)
    f32 = accum == "f32"
    rt = R.Raytracer(W, H, R.RT_ACCUM_F32 if f32 else R.RT_ACCUM_RGBA8)
    rt.set_scene(sp, mt)
    rt.enable_rng_capture(True)
    rt.set_camera(cam)
    for at in range(0, FRAMES, 256):
        rt.render_frames(tb[at:at + 256])
    rt.sync()
    got = rt.read_accum_f32() if f32 else rt.read_rgba8()
    rng = rt.read_rng_state()
    rt.close()
    want = O.OracleImage(W, H, O.ACCUM_F32 if f32 else 0)
    O.render(want, sp, mt, cam, 0, FRAMES)
    if f32:
        assert np.array_equal(got.view(np.uint32), want.accum.view(np.uint32))
    else:
        assert np.array_equal(got, want.rgba8)
    assert np.array_equal(rng, want.rng)
