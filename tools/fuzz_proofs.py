#!/usr/bin/env python3
"""Long-running soundness sweep for the proofs that let the path kernel SKIP exact tests (not a pytest test: run it on
the GPU box when one of the proofs changes; tests/test_gpu_parity.py holds a fixed sample of the same generators).

Every scene is rendered twice in the same arithmetic mode: with everything on (tile masks incl. the occlusion drop,
candidate lists, own-sphere drop, walk deferral, candidate queue, grid trace) and with RT_SPLIT=0 RT_SKY_TILES=0 RT_GRID=0 (every ray takes
the general trace against every sphere, no tile masks). The literal arithmetic is the same in both, so ANY differing
bit of the float image or of the final RNG states means a proof dropped a sphere that mattered.

usage: tools/fuzz_proofs.py [first_seed] [n_seeds] [--fast] [--big]
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import raytracer_amd as R
import scenes


FORCE_KIND = None  # --big: only the many-sphere scenes (candidate lists, queue, GRID trace)


def scene(rs):
    kind = rs.randint(0, 7)
    if FORCE_KIND is not None:
        kind = FORCE_KIND
    scale = float(rs.choice([0.01, 0.3, 1.0, 1.0, 7.0, 500.0]))
    if kind == 0:    # a dense bunch seen from close by or from inside: deep overlap of projections
        n = int(rs.choice([5, 12, 20, 32]))
        centers = (rs.random_sample((n, 3)) * 2 - 1) * 2.0 + np.array([0, 0, -4.0])
        radii = 0.3 + rs.random_sample(n) * 1.2
    elif kind == 1:  # spheres in a row behind each other, gaps from 0 to a radius
        n = int(rs.choice([3, 8, 24, 90]))
        gap = float(rs.choice([0.0, 1e-4, 0.05, 0.6]))
        centers = np.array([(rs.normal(0, 0.05), rs.normal(0, 0.05), -3.0 - gap * k) for k in range(n)])
        radii = 1.0 + rs.random_sample(n) * float(rs.choice([0.0, 1e-3, 0.3]))
    elif kind == 2:  # a big ground (mirror or diffuse by index) under a low camera, small spheres on it
        n = int(rs.choice([4, 9, 20]))
        big = float(rs.choice([30.0, 100.0, 1000.0, 20000.0]))
        h = float(rs.choice([0.01, 0.1, 0.5, 2.0]))
        centers = np.vstack([[0, -big - h, 0], (rs.random_sample((n - 1, 3)) * 2 - 1) * np.array([3, 0.3, 3]) + np.array([0, 0.2, -5.0])])
        radii = np.concatenate([[big], 0.2 + rs.random_sample(n - 1) * 0.6])
        if rs.randint(2):
            centers[[0, 1]] = centers[[1, 0]]; radii[[0, 1]] = radii[[1, 0]]
    elif kind == 3:  # nested and coincident spheres
        n = int(rs.choice([2, 5, 9]))
        centers = np.tile(np.array([[0.0, 0.0, -4.0]]), (n, 1)) + rs.normal(0, float(rs.choice([0.0, 1e-6, 0.02])), (n, 3))
        radii = np.sort(0.2 + rs.random_sample(n) * 1.5)[::-1].copy()
        if n > 2:
            radii[1] = radii[0]; centers[1] = centers[0]
    elif kind == 4:  # many spheres: candidate lists, the queue of the general trace, list overflow
        n = int(rs.choice([65, 100, 300, 1024]))
        spread = float(rs.choice([2.0, 8.0, 30.0]))
        centers = (rs.random_sample((n, 3)) * 2 - 1) * spread + np.array([0, 0, -1.5 * spread])
        radii = (0.2 + rs.random_sample(n)) * spread * float(rs.choice([0.02, 0.1, 0.3]))
    elif kind == 5:  # the camera inside a sphere / spheres of very different size
        n = int(rs.choice([3, 7, 33]))
        centers = (rs.random_sample((n, 3)) * 2 - 1) * 3.0 + np.array([0, 0, -3.0])
        radii = np.exp(rs.uniform(np.log(1e-3), np.log(8.0), n))
    else:            # zero, negative and power-of-two radii
        n = int(rs.choice([2, 6, 17]))
        centers = (rs.random_sample((n, 3)) * 2 - 1) * 2.5 + np.array([0, 0, -5.0])
        radii = rs.choice([0.0, -1.0, -0.3, 0.25, 0.5, 1.0, 2.0], n)
    centers = centers * scale
    radii = radii * scale
    mats = scenes.make_materials(rs.random_sample((3, 4)).astype(np.float32))
    spheres = scenes.make_spheres(centers.astype(np.float32), radii.astype(np.float32), rs.randint(0, 3, size=len(radii)))
    pos = (rs.random_sample(3) * 2 - 1) * scale * float(rs.choice([0.0, 0.0, 0.4, 2.0]))
    look = centers.mean(axis=0) - pos + (rs.random_sample(3) * 2 - 1) * 0.3 * scale
    if not np.isfinite(look).all() or np.linalg.norm(look) < 1e-6 * scale or abs(look[0]) + abs(look[2]) < 1e-6 * scale:
        look = np.array([0.1, 0.0, -1.0])
    cam = dict(pos=tuple(pos.tolist()), look=tuple(look.tolist()), fov=float(rs.choice([0.05, 0.1, 0.25, 0.45])))
    return kind, spheres, mats, cam


def render(W, H, flags, spheres, mats, cam, frames, env):
    for k in ("RT_SPLIT", "RT_SKY_TILES", "RT_GRID"):
        os.environ.pop(k, None)
    os.environ.update(env)
    rt = R.Raytracer(W, H, flags)
    rt.enable_rng_capture(True)
    rt.set_scene(spheres, mats)
    tb = np.array([0x3c23d70a + 77 * i for i in range(frames)], np.uint32)  # any time bits do
    rt.set_camera(cam)
    rt.render_frames(tb)
    rt.sync()
    out = (rt.read_accum_f32().view(np.uint32).copy(), rt.read_rng_state().copy(), rt.sky_tile_stats())
    rt.close()
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    first = int(args[0]) if args else 0
    n = int(args[1]) if len(args) > 1 else 200
    math = R.RT_MATH_FAST if "--fast" in sys.argv else R.RT_MATH_STRICT
    global FORCE_KIND
    if "--big" in sys.argv:
        FORCE_KIND = 4
    bad = 0
    t0 = time.time()
    by_kind = {}
    for seed in range(first, first + n):
        rs = np.random.RandomState(100000 + seed)
        kind, spheres, mats, camkw = scene(rs)
        W, H = (int(rs.choice([128, 256, 384])), int(rs.choice([96, 160, 224])))
        spp = int(rs.choice([1, 2, 4]))
        flags = R.RT_ACCUM_F32 | math | (R.RT_SHADE_CHECKER if seed % 3 == 0 else 0)
        cam = scenes.camera_raw(W, H, spp=spp, frames=2, **camkw)
        on = render(W, H, flags, spheres, mats, cam, 2, {"RT_GRID": "2"})  # 2: the grid trace also below the scene size from which it pays
        off = render(W, H, flags, spheres, mats, cam, 2, {"RT_SPLIT": "0", "RT_SKY_TILES": "0", "RT_GRID": "0"})
        same = np.array_equal(on[0], off[0]) and np.array_equal(on[1], off[1])
        if not same and math == R.RT_MATH_FAST and np.array_equal(on[1], off[1]):
            # fast arithmetic may turn a degenerate sphere (radius 0: 1 / r = inf) into NaN colours, and which NaN comes out of an operation
            # (sign, payload) depends on the operand order the compiler chose in each loop shape: NaN in both images counts as the same
            fa, fb = on[0].view(np.float32), off[0].view(np.float32)
            same = bool(np.all((on[0] == off[0]) | (np.isnan(fa) & np.isnan(fb))))
        k = by_kind.setdefault(kind, [0, 0, 0, 0])
        k[0] += 1; k[1] += 0 if same else 1; k[2] += on[2][0]; k[3] += on[2][1]
        if not same:
            bad += 1
            diff = int((on[1] != off[1]).sum())
            print(f"MISMATCH seed {seed} kind {kind}: {len(spheres)} spheres, {W}x{H} {spp} spp, {diff} pixels with another RNG state, camera {camkw}", flush=True)
        if (seed - first) % 50 == 49:
            print(f"... {seed - first + 1} scenes, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    for kind in sorted(by_kind):
        c = by_kind[kind]
        print(f"kind {kind}: {c[0]} scenes, {c[1]} mismatches, sky tiles {c[2]} of {c[3]}")
    print(f"{n} scenes from seed {first}, {'fast' if math else 'strict'}: {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
