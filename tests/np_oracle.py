"""Second, independently written restatement of the reference's compute pass — numpy float32, vectorised over pixels.

TEST INFRASTRUCTURE ONLY (like oracle/): a double-entry check of oracle/rt_oracle.c's FLOAT path. The reference ships
no vectors and cannot run here, so the C oracle rests on one transcription of the shader; this file is a second one,
written from the WGSL text (src/shaders/raytrace_kernal.wgsl, cited per function) and from DESIGN.md's written
arithmetic contract, not from the C file. Both must agree bit for bit (tests/test_oracle_double_entry.py).

What it proves: the two transcriptions read the shader the same way (operation order, which value feeds which, loop
bounds, the quirks) under the same stated contract. What it does not prove: that the contract is what a given
wgpu/naga/driver stack computes — WGSL leaves contraction and the precision of sqrt, division, length and normalize
open, so parity with the reference itself stays unpinned.

Contract: every + - * / sqrt is ONE correctly rounded binary32 operation (numpy float32 ufuncs are),
no contraction, dot(a, b) = (ax*bx + ay*by) + az*bz, length = sqrt(dot), normalize(v) = v / length(v),
mat*vec = ((c0*x + c1*y) + c2*z) + c3*w, rgba8unorm store = clamp to [0, 1] (NaN -> 0), * 255, round half to even,
load = q / 255.
"""
import numpy as np

F = np.float32
U = np.uint32

RAY_TMAX = F(10000000.0)  # wgsl:53
RAY_TMIN = F(0.001)       # wgsl:54
MAX_RAY_DEPTH = 10        # wgsl:55

ACCUM_F32, DISPATCH_FULL, SHADE_CHECKER, COMBINE_SUM = 0x1, 0x2, 0x4, 0x10


def _dot(ax, ay, az, bx, by, bz):
    return (ax * bx + ay * by) + az * bz


def _length_squared(x, y, z):
    """wgsl:306-309: let len = length(x); return len * len;"""
    ln = np.sqrt(_dot(x, y, z, x, y, z))
    return ln * ln


class _Rng:
    """wgsl:254-275, one u32 state per pixel; `step(mask)` is rand() for the pixels in mask."""

    def __init__(self, u, v, time_bits):
        with np.errstate(over="ignore"):
            x = (u << U(15)) ^ (u * U(1729) + U(8192374))
            y = (v << U(21)) ^ (v * U(271) + U(719827371))
            t = U(time_bits) * U(87189) + U(189273)
        self.state = (x ^ y ^ t).astype(U)
        self.calls = 0
        self.rand(np.ones(u.shape, bool))  # wgsl:262: one discarded rand()

    def rand(self, mask):
        s = self.state
        with np.errstate(over="ignore"):
            n = s * U(18197) + U(182489729)
        n = n ^ (n << U(15))
        n = n ^ (n >> U(22))
        n = n ^ (n << U(4))
        self.state = np.where(mask, n, s).astype(U)
        self.calls += int(mask.sum())
        bits = (self.state & U(0x007FFFFF)) | U(0x3F800000)
        return bits.view(F) - F(1.0)  # only meaningful where mask


def _mat_vec(m, x, y, z, w):
    """(camera.pixel_to_world * vec4).xyz, column-major (wgsl:87)"""
    return tuple(((m[0 + i] * x + m[4 + i] * y) + m[8 + i] * z) + m[12 + i] * w for i in range(3))


def compute_frame(width, height, flags, spheres, materials, camera, time_bits, image_rgba8, image_f32, rng_out):
    """One dispatch of `main` (wgsl:72-83) + the buffer->cache copy (raytracing.rs:195-199), in place.
    Returns the number of rand() calls. Arrays: image_rgba8 uint8 HxWx4, image_f32 float32 HxWx4 or None."""
    frame, total = int(camera["current_frame"]), int(camera["frames_to_render"])
    if frame >= total:  # wgsl:76
        return 0
    if flags & DISPATCH_FULL:
        cw_, ch_ = width, height
    else:  # raytracing.rs:192: dispatch_workgroups(w / 16, h / 16, 1) of 16 x 16 invocations
        cw_, ch_ = (width // 16) * 16, (height // 16) * 16
    if cw_ == 0 or ch_ == 0:
        return 0
    vv, uu = np.mgrid[0:ch_, 0:cw_]
    u, v = uu.reshape(-1).astype(U), vv.reshape(-1).astype(U)
    n = u.size
    m = np.asarray(camera["pixel_to_world"], F)
    pos = np.asarray(camera["pos"], F)
    spp = int(camera["samples_per_pixel"])
    zero, one = np.zeros(n, F), np.ones(n, F)

    rng = _Rng(u, v, time_bits)  # initialize_rng (wgsl:80)

    # send_rays (wgsl:114-123)
    def p2w(px, py):  # wgsl:85-88: f32(x), f32(size.y - y) with u32 wrap-around, 0, 1
        return _mat_vec(m, px.astype(F), (U(height) - py).astype(U).astype(F), F(0.0), F(1.0))

    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        cx, cy, cz = p2w(u, v)
        nx_, ny_, nz_ = p2w(u + U(1), v + U(1))
        dx_, dy_, dz_ = nx_ - cx, ny_ - cy, nz_ - cz  # get_pixel_delta (wgsl:90-94)
        colx, coly, colz = zero.copy(), zero.copy(), zero.copy()
        everyone = np.ones(n, bool)
        for _sample in range(spp):
            # get_random_ray (wgsl:174-180)
            px = F(-0.5) + rng.rand(everyone)
            py = F(-0.5) + rng.rand(everyone)
            rdx = cx + px * dx_
            rdy = cy + py * dy_
            rdz = cz + F(1.0) * dz_
            rox, roy, roz = pos[0] + zero, pos[1] + zero, pos[2] + zero
            # ray_color (wgsl:125-142)
            pcx, pcy, pcz = zero.copy(), zero.copy(), zero.copy()
            refl = one.copy()
            alive = everyone.copy()
            for _depth in range(MAX_RAY_DEPTH + 1):
                if not alive.any():
                    break
                # trace (wgsl:182-198)
                closest = np.full(n, RAY_TMAX, F)
                hit = np.zeros(n, bool)
                hidx = np.zeros(n, np.int64)
                ht = zero.copy()
                a = _length_squared(rdx, rdy, rdz)
                for i in range(len(spheres)):
                    sc, sr = np.asarray(spheres[i]["center"], F), F(spheres[i]["radius"])
                    # hit_sphere (wgsl:200-231)
                    ocx, ocy, ocz = rox - sc[0], roy - sc[1], roz - sc[2]
                    half_b = _dot(ocx, ocy, ocz, rdx, rdy, rdz)
                    c = _length_squared(ocx, ocy, ocz) - sr * sr
                    disc = half_b * half_b - a * c
                    ok = alive & ~(disc < F(0.0))
                    sq = np.sqrt(np.where(ok, disc, F(0.0)))
                    root = (-half_b - sq) / a
                    bad = (root <= RAY_TMIN) | (closest <= root)
                    root2 = (-half_b + sq) / a
                    root = np.where(bad, root2, root)
                    bad2 = bad & ((root2 <= RAY_TMIN) | (closest <= root2))
                    ok = ok & ~bad2
                    closest = np.where(ok, root, closest)
                    ht = np.where(ok, root, ht)
                    hidx = np.where(ok, i, hidx)
                    hit = hit | ok
                missed = alive & ~hit
                # miss (wgsl:144-148) and `color += miss(ray) * reflectance; break`
                if flags & SHADE_CHECKER:  # sky off
                    mx = my = mz = zero
                else:
                    t = F(0.5) * (F(1.0) - rdy)
                    mx = (F(1.0) - t) * F(1.0) + t * F(0.5)
                    my = (F(1.0) - t) * F(1.0) + t * F(0.7)
                    mz = (F(1.0) - t) * F(1.0) + t * F(1.0)
                pcx = np.where(missed, pcx + mx * refl, pcx)
                pcy = np.where(missed, pcy + my * refl, pcy)
                pcz = np.where(missed, pcz + mz * refl, pcz)
                alive = alive & hit
                if not alive.any():
                    break
                # hit record of the closest sphere (wgsl:225-228)
                centers = np.asarray(spheres["center"], F)[hidx]
                radius = np.asarray(spheres["radius"], F)[hidx]
                hpx, hpy, hpz = rox + ht * rdx, roy + ht * rdy, roz + ht * rdz  # ray_at (wgsl:233-235)
                nx = (hpx - centers[:, 0]) / radius
                ny = (hpy - centers[:, 1]) / radius
                nz = (hpz - centers[:, 2]) / radius
                # scatter (wgsl:150-168)
                mcol = np.asarray(materials["color"], F)[np.asarray(spheres["material_index"])[hidx]]
                local_reflectance = F(0.90)
                if flags & SHADE_CHECKER:
                    # emission material.color * reflectance where sin(4 pi n.x) * cos(4 pi n.y) > 0 (the dead
                    # terms of wgsl:157-158), stated as: floor(4 n.x) + floor(4 n.y + 1/2) is even
                    s = np.floor(F(4.0) * nx) + np.floor(F(4.0) * ny + F(0.5))
                    half = s * F(0.5)
                    k = np.where(half == np.floor(half), refl, F(0.0))
                else:
                    k = (F(1.0) - local_reflectance) * refl
                pcx = np.where(alive, pcx + mcol[:, 0] * k, pcx)
                pcy = np.where(alive, pcy + mcol[:, 1] * k, pcy)
                pcz = np.where(alive, pcz + mcol[:, 2] * k, pcz)
                refl = np.where(alive, refl * local_reflectance, refl)
                rox, roy, roz = np.where(alive, hpx, rox), np.where(alive, hpy, roy), np.where(alive, hpz, roz)
                mirror = alive & (hidx % 2 == 0)
                diffuse = alive & ~mirror
                # reflection (wgsl:170-172)
                k2 = F(2.0) * _dot(rdx, rdy, rdz, nx, ny, nz)
                mdx, mdy, mdz = rdx - k2 * nx, rdy - k2 * ny, rdz - k2 * nz
                # random_unit_vector (wgsl:281-294): each pixel draws until ITS point is inside the unit sphere
                qx, qy, qz = zero.copy(), zero.copy(), zero.copy()
                need = diffuse.copy()
                while need.any():
                    r0 = rng.rand(need)
                    r1 = rng.rand(need)
                    r2 = rng.rand(need)
                    tx, ty, tz = r0 * F(2.0) - F(1.0), r1 * F(2.0) - F(1.0), r2 * F(2.0) - F(1.0)
                    acc = need & (_length_squared(tx, ty, tz) < F(1.0))
                    qx, qy, qz = np.where(acc, tx, qx), np.where(acc, ty, qy), np.where(acc, tz, qz)
                    need = need & ~acc
                ql = np.sqrt(_dot(qx, qy, qz, qx, qy, qz))
                ddx, ddy, ddz = nx + qx / ql, ny + qy / ql, nz + qz / ql
                rdx = np.where(mirror, mdx, np.where(diffuse, ddx, rdx))
                rdy = np.where(mirror, mdy, np.where(diffuse, ddy, rdy))
                rdz = np.where(mirror, mdz, np.where(diffuse, ddz, rdz))
            colx, coly, colz = colx + pcx, coly + pcy, colz + pcz  # wgsl:120
        sppf = F(spp)
        p = np.stack([colx / sppf, coly / sppf, colz / sppf, one], axis=1)  # wgsl:122, to_vec4 (wgsl:311-313)

        # store_color / combine_pixel_cache_color (wgsl:96-112); color_buffer and color_cache are Rgba8Unorm
        # (raytracing.rs:30,47); RT_ACCUM_F32 keeps the un-quantised float image instead
        vi, ui = v.astype(np.int64), u.astype(np.int64)
        if flags & ACCUM_F32:
            cached = image_f32[vi, ui]
        else:
            cached = image_rgba8[vi, ui].astype(F) / F(255.0)
        if flags & COMBINE_SUM:
            out = cached + p
        else:
            cw = F(frame) / F(frame + 1)
            rw = F(1.0) - cw
            out = p * rw + cached * cw
        if flags & ACCUM_F32:
            image_f32[vi, ui] = out
        q = np.rint(np.clip(np.where(np.isnan(out), F(0.0), out), F(0.0), F(1.0)) * F(255.0)).astype(np.uint8)
        image_rgba8[vi, ui] = q
    rng_out[vi, ui] = rng.state
    return rng.calls


def render(width, height, flags, spheres, materials, camera, n_frames):
    """App loop of a fresh process (app.rs:93-116): time += 0.01f before every redraw (time.rs:72-77), compute, then
    current_frame += 1 (camera.rs:253). Returns (rgba8, f32 or None, rng state, rand calls)."""
    rgba8 = np.zeros((height, width, 4), np.uint8)
    f32 = np.zeros((height, width, 4), F) if flags & ACCUM_F32 else None
    rng = np.zeros((height, width), U)
    cam = np.array(camera, copy=True)
    t = F(0.0)
    calls = 0
    for _ in range(n_frames):
        t = F(t + F(0.01))
        calls += compute_frame(width, height, flags, spheres, materials, cam, int(np.array(t, F).view(U)), rgba8, f32, rng)
        cam["current_frame"] += 1
    return rgba8, f32, rng, calls
