"""Test-side builders of the reference's inputs (independent of the product's C++ host mirror).

Byte layouts follow reference src/resources/objects.rs:17-24, material.rs:5-13, camera.rs:38-48.
The camera is derived in float64 and rounded once to f32, so it cross-checks the host mirror's
f32 matrix code to ~1e-6 instead of sharing its rounding.
"""
import numpy as np

from oracle_lib import CAMERA_DTYPE, MATERIAL_DTYPE, SPHERE_DTYPE


def make_spheres(centers, radii, mats):
    n = len(radii)
    s = np.zeros(n, SPHERE_DTYPE)
    s["center"] = np.asarray(centers, np.float32).reshape(n, 3)
    s["padding"] = s["center"][:, 2].copy().view(np.uint32)  # center.xyzz (objects.rs:44)
    s["radius"] = np.asarray(radii, np.float32)
    s["material_index"] = np.asarray(mats, np.uint32)
    return s


def make_materials(colors4, rest3=None):
    n = len(colors4)
    m = np.zeros(n, MATERIAL_DTYPE)
    m["color"] = np.asarray(colors4, np.float32).reshape(n, 4)
    if rest3 is not None:
        rest3 = np.asarray(rest3, np.float32).reshape(n, 3)
        m["metallic"], m["roughness"], m["specular"] = rest3[:, 0], rest3[:, 1], rest3[:, 2]
    return m


def default_scene(seed=1):
    """world.rs:34-40: four spheres, three uniform-[0,1) random materials (seeded here)."""
    rs = np.random.RandomState(seed)
    mats = make_materials(rs.random_sample((3, 4)).astype(np.float32), rs.random_sample((3, 3)).astype(np.float32))
    spheres = make_spheres(
        [(0.0, -100.5, 0.0), (-1.0, 0.0, -1.0), (1.0, 0.0, -1.0), (0.0, 0.0, -1.0)], [100.0, 0.5, 0.5, 0.5], [0, 1, 2, 0]
    )
    return spheres, mats


def random_bunch(amount, n_materials, seed):
    """objects.rs:27-40 geometry with in-range material indices."""
    rs = np.random.RandomState(seed)
    root = np.float32(np.sqrt(np.float32(amount)))
    lo = np.array([-1.0, -1.0, -1.0], np.float32) * root
    hi = np.array([1.0, 1.0, -2.0], np.float32) * root
    r = rs.random_sample((amount, 3)).astype(np.float32)
    centers = r * (hi - lo) + lo
    mats = make_materials(rs.random_sample((n_materials, 4)).astype(np.float32))
    idx = rs.randint(0, n_materials, size=amount)
    return make_spheres(centers, np.ones(amount, np.float32), idx), mats


def camera_raw(width, height, spp=1, frames=8, current_frame=0, pos=(0.0, 0.0, 0.0), look=(0.0, 0.0, -1.0),
               fov=0.25, z_near=1.0, z_far=100000.0):
    """camera.rs:77-131 in float64, rounded to f32 at the end."""
    u, v = float(width), float(height)
    clip_to_pixel = np.array([[u / 2, 0, 0, u / 2], [0, v / 2, 0, v / 2], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)
    n, f = z_near, z_far
    t = np.tan(np.float64(np.float32(fov)) * np.pi / 2.0)
    a = u / v
    # from_cols_array: the 16 numbers are COLUMNS (camera.rs:106-111)
    proj = np.array([[1 / (a * t), 0, 0, 0], [0, 1 / t, 0, 0], [0, 0, -f / (f - n), -1.0], [0, 0, -f * n, 0]], np.float64).T
    # Mat4::look_to_rh(eye=0, dir, up)
    d = np.asarray(look, np.float64)
    fwd = d / np.linalg.norm(d)
    s = np.cross(fwd, [0.0, 1.0, 0.0]); s /= np.linalg.norm(s)
    upv = np.cross(s, fwd)
    rot = np.array([[s[0], s[1], s[2], 0], [upv[0], upv[1], upv[2], 0], [-fwd[0], -fwd[1], -fwd[2], 0], [0, 0, 0, 1]], np.float64)
    tr = np.eye(4)
    tr[:3, 3] = -np.asarray(pos, np.float64)
    w2p = clip_to_pixel @ proj @ rot @ tr
    p2w = np.linalg.inv(w2p)
    c = np.zeros((), CAMERA_DTYPE)
    c["world_to_pixel"] = w2p.T.reshape(16).astype(np.float32)  # column-major
    c["pixel_to_world"] = p2w.T.reshape(16).astype(np.float32)
    c["pos"] = np.array([pos[0], pos[1], pos[2], pos[2]], np.float32)
    c["focal_length"] = z_near
    c["samples_per_pixel"], c["frames_to_render"], c["current_frame"] = spp, frames, current_frame
    return c
