"""Grid trace against the loop over all spheres, by scene size: random_bunch(n) (objects.rs:27-40) at 1080p, 4 spp, one pass per
launch x 8, strict and fast. Scratch probe for the rule that switches the grid on (rt_api.cpp, build_grid), not the bench."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import raytracer_amd as R
import oracle_lib as O, scenes


def timed(sp, mt, flags, grid, reps=8, W=1920, H=1080, spp=4):
    os.environ["RT_GRID"] = grid
    rt = R.Raytracer(W, H, flags)
    rt.set_scene(sp, mt)
    cam = scenes.camera_raw(W, H, spp=spp, frames=1000)
    tb = O.time_bits_table(reps + 3)
    for f in range(2):
        cam["current_frame"] = f; rt.set_camera(cam); rt.set_time(int(tb[f])); rt.compute(); rt.sync()
    rt.enable_timing(True)
    for i in range(reps):
        cam["current_frame"] = i + 2
        rt.set_camera(cam); rt.set_time(int(tb[i + 2])); rt.compute()
    ms, n = rt.timing()
    rt.close()
    return ms / reps


if __name__ == "__main__":
    print("| spheres | strict: all spheres | strict: grid | fast: all spheres | fast: grid |")
    print("|---|---|---|---|---|")
    for n in (65, 100, 150, 200, 300, 500, 1024, 4096, 16384):
        sp, mt = scenes.random_bunch(n, 3, 7)
        row = []
        for math in (R.RT_MATH_STRICT, R.RT_MATH_FAST):
            for grid in ("0", "1"):
                row.append(timed(sp, mt, R.RT_ACCUM_F32 | math, grid, reps=8 if n <= 4096 else 3))
        print("| %d | %.3f ms | %.3f ms | %.3f ms | %.3f ms |" % (n, *row), flush=True)
