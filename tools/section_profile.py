"""Per-section share of wave time in rt_path_kernel. Needs a library built with -DRT_PROFILE_SECTIONS=1
(tools/section_profile.sh swaps it in on the GPU box). The path kernel then reports shader-clock cycles per
section through the counter slots: T=[G] draws, D=trace prelude, H=phase 1, R=phase 2, M=shade, S=wave
lifetime, Sm=loop iterations, J=lifetime in 100 MHz ticks."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import raytracer_amd as R
import oracle_lib as O, scenes

def run(name, W, H, spp, scene, shade, math):
    sp, mt = scene
    rt = R.Raytracer(W, H, R.RT_ACCUM_F32 | shade | math)
    rt.set_scene(sp, mt)
    cam = scenes.camera_raw(W, H, spp=spp, frames=1000)
    tb = O.time_bits_table(4)
    for i in range(3):  # third launch is measured (tile order settled)
        rt.reset_counters()
        cam["current_frame"] = i
        rt.set_camera(cam); rt.set_time(int(tb[i])); rt.compute(); rt.sync()
    c = rt.counters()
    life = c["S"]; iters = c["Sm"]
    parts = [("[G] draws", c["T"]), ("trace prelude", c["D"]), ("phase 1", c["H"]), ("phase 2", c["R"]), ("shade", c["M"])]
    other = life - sum(v for _, v in parts)
    print(f"{name} {'fast' if math else 'strict'}: {iters} wave-iterations, {life/iters:.0f} wave-cycles per iteration, "
          f"shader clock {life / max(c['J'], 1) * 0.1:.2f} GHz, mean wave lifetime {c['J'] / 6144 * 1e-5:.2f} ms")
    t0 = (~c["Sd"]) & 0xffffffffffffffff
    span = (c["N"] - t0) * 1e-5
    occ = rt.occupancy()
    p2_lanes, p2_waves = occ["slots_trace"], occ["slots_reject"]
    split = os.environ.get("RT_SPLIT", "1") != "0" and sp.shape[0] <= 32  # SPLIT loop shape: "phase 2" = the one candidate walk, "shade" = the one scatter
    n = 6144  # 24 waves per CU x 256 CUs
    mean_start = (c["P"] / n - (t0 & 0xffffffff)) * 1e-5
    mean_end = (c["rand_calls"] / n - (t0 & 0xffffffff)) * 1e-5
    print(f"   launch span {span:.2f} ms; mean wave start +{mean_start:.2f} ms, mean wave end +{mean_end:.2f} ms")
    print(f"   {'SPLIT shape; ' if split else ''}candidate walk: {p2_waves / iters:.2f} wave-iterations per loop iteration, {p2_lanes / max(p2_waves, 1):.1f} of 64 lanes active, "
          f"{p2_lanes / iters:.1f} candidates per wave and loop iteration")
    other = life - sum(v for _, v in parts)
    for k, v in parts + [("[E][F][A] + loop", other)]:
        print(f"   {k:18s} {100.0*v/life:5.1f} %   {v/iters:7.0f} cycles/iteration")
    rt.close()

if __name__ == "__main__":
    for math in (R.RT_MATH_STRICT, R.RT_MATH_FAST):
        run("C3", 3840, 2160, 64, scenes.random_bunch(20, 3, 3), R.RT_SHADE_CHECKER, math)
        run("C4", 1920, 1080, 4, scenes.random_bunch(1024, 16, 4), 0, math)
        run("C2", 1920, 1080, 16, scenes.default_scene(1), 0, math)
