"""One seed of tools/fuzz_proofs.py looked at closely: which switch makes the difference, how many pixels, how large.
usage: tools/fuzz_one.py <seed> [--fast]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import raytracer_amd as R
import scenes
import fuzz_proofs as F

seed = int(sys.argv[1])
math = R.RT_MATH_FAST if "--fast" in sys.argv else R.RT_MATH_STRICT
rs = np.random.RandomState(100000 + seed)
kind, spheres, mats, camkw = F.scene(rs)
W, H = (int(rs.choice([128, 256, 384])), int(rs.choice([96, 160, 224])))
spp = int(rs.choice([1, 2, 4]))
flags = R.RT_ACCUM_F32 | math | (R.RT_SHADE_CHECKER if seed % 3 == 0 else 0)
cam = scenes.camera_raw(W, H, spp=spp, frames=2, **camkw)
print("kind", kind, len(spheres), "spheres", W, H, spp, "checker" if seed % 3 == 0 else "")
print(spheres)
ref = F.render(W, H, flags, spheres, mats, cam, 2, {"RT_SPLIT": "0", "RT_SKY_TILES": "0", "RT_GRID": "0"})
strict = F.render(W, H, (flags & ~R.RT_MATH_FAST) | R.RT_MATH_STRICT, spheres, mats, cam, 2, {})
for name, env in (("all on", {"RT_GRID": "2"}), ("RT_SKY_TILES=0 (no masks; SPLIT needs them)", {"RT_SKY_TILES": "0"}), ("RT_SPLIT=0 (masks for sky tiles only)", {"RT_SPLIT": "0"})):
    got = F.render(W, H, flags, spheres, mats, cam, 2, env)
    a, b = got[0].view(np.float32).reshape(H, W, 4), ref[0].view(np.float32).reshape(H, W, 4)
    ua, ub = got[0].view(np.uint32).reshape(H, W, 4), ref[0].view(np.uint32).reshape(H, W, 4)
    ys, xs = np.nonzero((ua != ub).any(axis=2))
    both_nan = int(sum(1 for x, y in zip(xs, ys) if np.isnan(a[y, x]).any() and np.isnan(b[y, x]).any()))
    finite = [float(np.abs(a[y, x] - b[y, x]).max()) for x, y in zip(xs, ys) if not (np.isnan(a[y, x]).any() or np.isnan(b[y, x]).any())]
    print(f"{name}: {len(ys)} pixels differ in bits from the trace of everything ({both_nan} of them NaN in both, max abs of the others {max(finite) if finite else 0:.3g}), "
          f"RNG states differ in {(got[1] != ref[1]).sum()}")
    for x, y in list(zip(xs, ys))[:3]:
        print("   pixel", (int(x), int(y)), [hex(v) for v in ua[y, x]], [hex(v) for v in ub[y, x]])
s = strict[0].view(np.float32).reshape(H, W, 4); b = ref[0].view(np.float32).reshape(H, W, 4)
print("fast (trace of everything) against strict: max abs", np.nanmax(np.abs(s - b)), "NaN pixels", int(np.isnan(b).any(axis=2).sum()), "/", int(np.isnan(s).any(axis=2).sum()), "pixels with another RNG state", (strict[1] != ref[1]).sum())
