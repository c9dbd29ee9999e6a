"""Cost of the per-pixel parts ([F] frame end, [A] pixel hand-out): same view and sample count rendered as
many pixels x few samples or few pixels x many samples. Scratch probe."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import raytracer_amd as R
import oracle_lib as O, scenes
from quick_perf import run

if __name__ == "__main__":
    sc = scenes.random_bunch(20, 3, 3)
    for math in (R.RT_MATH_STRICT, R.RT_MATH_FAST):
        for (W, H, spp) in ((7680, 4320, 16), (3840, 2160, 64), (1920, 1080, 256), (960, 540, 1024)):
            run(f"{W}x{H} {spp}spp checker20", W, H, spp, sc, R.RT_SHADE_CHECKER, math)
