"""CPU tests of the headless C++ host mirror (librt_host.so): Camera, World, Time, PNG."""
import os

import numpy as np

import oracle_lib as O
import scenes
from raytracer_amd import host


def test_camera_matches_closed_form_and_float64():
    for (w, h) in [(640, 360), (1920, 1080), (3840, 2160), (2048, 1024)]:
        c = host.camera_raw(w, h, spp=3, frames=9, current_frame=2)
        assert (int(c["samples_per_pixel"]), int(c["frames_to_render"]), int(c["current_frame"])) == (3, 9, 2)
        assert c["focal_length"] == 1.0 and not c["pos"].any()
        # p(x, y') = (a t (2x/W - 1), t (2y'/H - 1), -1), no w-divide
        m = c["pixel_to_world"].reshape(4, 4).T.astype(np.float64)
        t, a = np.tan(np.pi / 8), w / h
        for (x, y) in [(0, 0), (w, h), (w / 2, h / 2), (17, 5)]:
            p = m @ np.array([x, y, 0.0, 1.0])
            want = np.array([a * t * (2 * x / w - 1), t * (2 * y / h - 1), -1.0])
            assert np.abs(p[:3] - want).max() < 2e-6
        ref = scenes.camera_raw(w, h, spp=3, frames=9, current_frame=2)
        assert np.abs(c["pixel_to_world"] - ref["pixel_to_world"]).max() < 1e-6
        assert np.abs(c["world_to_pixel"] - ref["world_to_pixel"]).max() < 1e-3 * max(w, h) / 640


def test_moved_camera_against_float64():
    c = host.camera_raw(800, 600, pos=(0.3, 0.2, 1.5), look=(0.2, -0.1, -1.0), fov=0.3)
    ref = scenes.camera_raw(800, 600, pos=(0.3, 0.2, 1.5), look=(0.2, -0.1, -1.0), fov=0.3)
    assert np.abs(c["pixel_to_world"] - ref["pixel_to_world"]).max() < 2e-6
    assert np.allclose(c["pos"], [0.3, 0.2, 1.5, 1.5])
    # inverse really inverts
    w2p = c["world_to_pixel"].reshape(4, 4).T.astype(np.float64)
    p2w = c["pixel_to_world"].reshape(4, 4).T.astype(np.float64)
    assert np.abs(w2p @ p2w - np.eye(4)).max() < 1e-3


def test_time_table_is_repeated_f32_addition():
    t = host.time_table(256)
    assert np.array_equal(t, O.time_bits_table(256))
    assert (int(t[0]), int(t[4]), int(t[63]), int(t[255])) == (0x3C23D70A, 0x3D4CCCCC, 0x3F23D705, 0x4023D702)
    # continuing from a later start equals one long run (time is never reset, camera.rs:274-277)
    assert np.array_equal(host.time_table(6, int(t[9])), t[10:16])


def test_default_scene_layout():
    s, m = host.scene_default(1)
    assert np.array_equal(s["center"], np.array([[0, -100.5, 0], [-1, 0, -1], [1, 0, -1], [0, 0, -1]], np.float32))
    assert np.array_equal(s["radius"], np.array([100, 0.5, 0.5, 0.5], np.float32))
    assert list(s["material_index"]) == [0, 1, 2, 0]
    assert np.array_equal(s["padding"], s["center"][:, 2].view(np.uint32))  # center.xyzz (objects.rs:44)
    assert ((m["color"] >= 0) & (m["color"] < 1)).all() and (m["padding"] == 0).all()
    s2, m2 = host.scene_default(1)
    assert s.tobytes() == s2.tobytes() and m.tobytes() == m2.tobytes()  # seeded: reproducible
    assert m.tobytes() != host.scene_default(2)[1].tobytes()


def test_random_bunch_geometry():
    n = 1024
    s, m = host.scene_random_bunch(n, 16, 4)
    root = np.sqrt(np.float32(n))
    c = s["center"]
    assert (c[:, :2] >= -root).all() and (c[:, :2] < root).all()
    assert (c[:, 2] > -2 * root - 1e-3).all() and (c[:, 2] <= -root + 1e-3).all()  # z in (-2 sqrt n, -sqrt n]
    assert (s["radius"] == 1.0).all() and s["material_index"].max() < 16 and len(set(s["material_index"])) == 16
    assert m.shape == (16,)


def test_png_writer_round_trip(tmp_path):
    from PIL import Image

    rs = np.random.RandomState(3)
    img = rs.randint(0, 256, (37, 53, 4)).astype(np.uint8)
    p = os.path.join(tmp_path, "a.png")
    host.write_png(p, img)
    back = np.asarray(Image.open(p).convert("RGB"))
    assert np.array_equal(back, img[..., :3])
    host.write_png(p, img, present=True)  # pow(c, 2.2) then sRGB encode: close to identity, monotone
    shown = np.asarray(Image.open(p).convert("RGB")).astype(int)
    lin = (img[..., :3] / 255.0) ** 2.2
    srgb = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * lin ** (1 / 2.4) - 0.055)
    assert np.abs(shown - np.rint(srgb * 255)).max() <= 1


def test_unorm8_load_formula_is_exact_for_all_256_values():
    """The kernel's rgba8unorm load, fma(q, c, RN(q * c_lo)) with c = f32(1/255), c_lo = f32(1/255 - c)
    (rt_kernels.hip unorm8_to_f32), equals the correctly rounded q / 255 for every q; the constants in
    the source are these."""
    import re, os

    c = np.float32(1) / np.float32(255)
    c_lo = np.float32(1 / 255 - float(c))
    q = np.arange(256, dtype=np.float32)
    t = (q * c_lo).astype(np.float64)  # RN_f32(q * c_lo)
    fma = (q.astype(np.float64) * float(c) + t).astype(np.float32)  # exact product and sum in f64, one rounding
    assert np.array_equal(fma, q / np.float32(255))
    src = open(os.path.join(os.path.dirname(__file__), "..", "raytracer_amd", "csrc", "rt_kernels.hip")).read()
    m = re.search(r"__builtin_fmaf\(f, (\S+)f, f \* (\S+)f\)", src)
    assert float.fromhex(m.group(1)) == float(c) and float.fromhex(m.group(2)) == float(c_lo)


def test_camera_controls_match_the_reference_rules():
    cam = host.Camera(640, 360)
    raw, fov = cam.raw()
    assert (int(raw["samples_per_pixel"]), int(raw["frames_to_render"]), fov) == (1, 8, 0.25)
    for _ in range(10):
        cam.key("K"); cam.key("I")  # x2, clamped to 256 (camera.rs:136,138)
    raw, _ = cam.raw()
    assert (int(raw["samples_per_pixel"]), int(raw["frames_to_render"])) == (256, 256)
    cam.key("J"); cam.key("U")
    raw, _ = cam.raw()
    assert (int(raw["samples_per_pixel"]), int(raw["frames_to_render"])) == (128, 128)
    cam.key("Backspace")
    raw, _ = cam.raw()
    assert (int(raw["samples_per_pixel"]), int(raw["frames_to_render"])) == (1, 1)
    # frame counter: render_finished after frames_to_render increases; any control resets it (camera.rs:274-288)
    assert not cam.render_finished()
    cam.increase_frame()
    assert cam.render_finished()
    cam.key("W")
    assert not cam.render_finished()
    raw, _ = cam.raw()
    assert np.allclose(raw["pos"], [0, 0, -0.05, -0.05], atol=1e-7)  # forward = -z, speed 0.05
    cam.key("D"); cam.key("Space")
    raw, _ = cam.raw()
    assert np.allclose(raw["pos"][:3], [0.05, 0.05, -0.05], atol=1e-7)
    # zoom: fov' = 2 atan(exp(ln tan(fov pi/2) + val/5)) / pi (camera.rs:193-197)
    cam.zoom(1.0)
    _, fov = cam.raw()
    want = 2 * np.arctan(np.exp(np.log(np.tan(0.25 * np.pi / 2)) + 0.2)) / np.pi
    assert abs(fov - want) < 1e-6
    # looking at the centre pixel keeps the view direction; off-centre turns towards that pixel
    c2 = host.Camera(640, 360)
    before, _ = c2.raw()
    c2.look_at_pixel(0.0, 0.0)
    after, _ = c2.raw()
    assert np.abs(before["pixel_to_world"] - after["pixel_to_world"]).max() < 1e-6
    c2.look_at_pixel(100.0, 0.0)
    turned, _ = c2.raw()
    m = turned["pixel_to_world"].reshape(4, 4).T.astype(np.float64)
    centre = (m @ np.array([320.0, 180.0, 0.0, 1.0]))[:3]
    assert centre[0] > 0.05 and centre[2] < 0  # now looking to the right, still forward
    ref = scenes.camera_raw(640, 360, look=centre / np.linalg.norm(centre))
    assert np.abs(turned["pixel_to_world"] - ref["pixel_to_world"]).max() < 1e-5


def _ckpt_bytes(width, height, flags, frame=3, image=None, magic=b"RTCKPT01"):
    """A checkpoint file written by hand from the documented layout (include/rt_host.h)."""
    import struct
    n = width * height * (16 if flags & 1 else 4)
    head = magic + struct.pack("<10I2Q10f", width, height, flags, frame, 8, 2, 0x3CF5C28F, 4, 3, 0, 0x1234, n,
                               0, 0, 0, 0, 0, -1, 1.0, 1e5, 0.25, 0.05)
    head += b"\0" * (128 - len(head))
    return head + (image if image is not None else b"\x07" * n)


def test_checkpoint_header_is_parsed_and_validated_without_a_gpu(tmp_path):
    p = tmp_path / "a.ckpt"
    p.write_bytes(_ckpt_bytes(32, 16, 1))
    info = host.checkpoint_info(str(p))
    assert (info["width"], info["height"], info["flags"], info["current_frame"]) == (32, 16, 1, 3)
    assert info["time_bits"] == 0x3CF5C28F and info["scene_hash"] == 0x1234 and info["image_bytes"] == 32 * 16 * 16
    for bad, why in [(_ckpt_bytes(32, 16, 1)[:100], "shorter"), (_ckpt_bytes(32, 16, 1, magic=b"NOTACKPT"), "magic"),
                     (_ckpt_bytes(32, 16, 1)[:-5], "truncated"), (_ckpt_bytes(32, 16, 0, image=b"\0" * 7), "truncated")]:
        p.write_bytes(bad)
        try:
            host.checkpoint_info(str(p))
        except Exception as e:
            assert why in str(e), (why, str(e))
        else:
            raise AssertionError(f"accepted a checkpoint that is {why}")
