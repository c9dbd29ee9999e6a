"""Marginal cost of one phase-1 sphere test: the C3 scene padded with far-away spheres that no ray
hits (same image, same rays), timed at several sphere counts. Scratch probe, not the bench."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import raytracer_amd as R
import oracle_lib as O, scenes

def timed(W, H, spp, sp, mt, flags, reps=3):
    rt = R.Raytracer(W, H, flags)
    rt.set_scene(sp, mt)
    cam = scenes.camera_raw(W, H, spp=spp, frames=1000)
    tb = O.time_bits_table(reps + 3)
    rt.enable_counters(True)
    rt.set_camera(cam); rt.set_time(int(tb[0])); rt.compute(); rt.sync()
    c = rt.counters(); rt.enable_counters(False)
    cam["current_frame"] = 1; rt.set_camera(cam); rt.set_time(int(tb[1])); rt.compute(); rt.sync()
    rt.enable_timing(True)
    for i in range(reps):
        cam["current_frame"] = i + 2
        rt.set_camera(cam); rt.set_time(int(tb[i + 2])); rt.compute()
    ms, n = rt.timing()
    rt.close()
    return ms / n, c

if __name__ == "__main__":
    W, H, spp = 3840, 2160, 64
    sp0, mt = scenes.random_bunch(20, 3, 3)
    for math in (R.RT_MATH_STRICT, R.RT_MATH_FAST):
        base = None
        for n in (20, 32, 52, 64):
            sp = np.zeros(n, dtype=sp0.dtype)
            sp[:20] = sp0
            for i in range(20, n):
                sp[i]["center"] = (1.0e4 + 10.0 * i, 1.0e4, 1.0e4)
                sp[i]["radius"] = 0.01
                sp[i]["material_index"] = 0
            ms, c = timed(W, H, spp, sp, mt, R.RT_ACCUM_F32 | R.RT_SHADE_CHECKER | math)
            wave_rays = c["R"] / 64.0
            if base is None: base = (n, ms)
            marg = (ms - base[1]) / max(n - base[0], 1) * 1e6 / wave_rays * 1024  # ns of one SIMD per wave-ray-sphere
            print(f"{'fast' if math else 'strict'} n={n:3d} {ms:8.3f} ms  rays {c['R']}  D/ray {c['D']/c['R']:.3f}  marginal {marg:7.3f} SIMD-ns per wave-ray sphere test "
                  f"(= {marg*2.1:5.2f} cycles @2.1GHz)", flush=True)
