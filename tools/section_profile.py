"""Per-section share of wave time in rt_path_kernel. Needs a library built with -DRT_PROFILE_SECTIONS=1
(tools/section_profile.sh swaps it in on the GPU box). The path kernel then reports shader-clock cycles per
section through the counter slots: T=[G] draws, D=trace prelude, H=phase 1, R=phase 2, M=shade, S=wave
lifetime, Sm=loop iterations, J=lifetime in 100 MHz ticks."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import raytracer_amd as R
import oracle_lib as O, scenes
from raytracer_amd import configs

def bench_scene(name):
    c = configs.get(name)
    return c["spheres"], c["materials"]

def run(name, W, H, spp, scene, shade, math):
    sp, mt = scene
    rt = R.Raytracer(W, H, R.RT_ACCUM_F32 | shade | math)
    rt.set_scene(sp, mt)
    cam = scenes.camera_raw(W, H, spp=spp, frames=1000)
    batch = int(os.environ.get("SECTION_BATCH", "1"))  # passes per launch (the bench's launches hold 20)
    tb = O.time_bits_table(3 * batch + 1)
    for i in range(3):  # third launch is measured (tile order settled)
        rt.reset_counters()
        cam["current_frame"] = i * batch
        rt.set_camera(cam)
        if batch > 1:
            rt.render_frames(tb[i * batch:(i + 1) * batch])
        else:
            rt.set_time(int(tb[i])); rt.compute()
        rt.sync()
    if batch > 1:
        name = f"{name} ({batch}-pass launch)"
    c = rt.counters()
    life = c["S"]; iters = c["Sm"]
    only = os.environ.get("SECTION_ONLY")
    if only is not None:  # library built with -DRT_PROFILE_ONLY=<only>: that section's share with two clock reads per iteration
        label, key = [("[G] draws", "T"), ("trace prelude", "D"), ("phase 1", "H"), ("phase 2", "R"), ("shade", "M")][int(only)]
        print(f"ONLY {name} {'fast' if math else 'strict'} | {label} | {100.0 * c[key] / life:.2f} | {c[key] / iters:.0f} | {life / iters:.0f}")
        rt.close()
        return
    if os.environ.get("SECTION_LANES") == "hist":  # -DRT_PROFILE_SECTIONS=3: candidates per ray at the walk
        prim = [c[k] for k in ("T", "D", "H", "R", "M")]; bounce = [c[k] for k in ("Sd", "N", "P", "rand_calls")]
        print(f"{name} {'fast' if math else 'strict'}: primary rays with 1, 2, 3, 4, 5+ candidates: " + ", ".join(f"{100.0 * v / max(sum(prim), 1):.1f} %" for v in prim)
              + f" (mean {sum((i + 1) * v for i, v in enumerate(prim)) / max(sum(prim), 1):.2f}); bounce rays with 1, 2, 3, 4+: "
              + ", ".join(f"{100.0 * v / max(sum(bounce), 1):.1f} %" for v in bounce) + f"; primary : bounce = {sum(prim) / max(sum(bounce), 1):.2f}")
        rt.close()
        return
    if os.environ.get("SECTION_LANES"):  # library built with -DRT_PROFILE_SECTIONS=2: lane occupancy per section
        occ = rt.occupancy()
        print(f"{name} {'fast' if math else 'strict'}: {iters} wave-iterations")
        labels = ("[G0] new ray", "[P1] prelude + phase 1", "rejection round", "scatter", "normalize")
        if len(sp) > 64:  # LIST shape: the slots of the rounds and of normalize count the grid walk instead
            labels = ("[G0] new ray", "[C] general trace", "grid trip: lanes still walking", "exact test of a tile-list entry", "grid trip: lanes taking a list entry")
        for label, key in zip(labels, ("T", "D", "H", "R", "M")):
            lanes, execs = c[key] & ((1 << 38) - 1), (c[key] >> 38) * 16  # (the kernel reports executions / 16 per wave)
            print(f"   {label:36s} {execs / iters:6.2f} executions per iteration, {lanes / max(execs, 1):5.1f} of 64 lanes")
        print(f"   {'candidate walk trip':36s} {occ['slots_reject'] / iters:6.2f} executions per iteration, {occ['slots_trace'] / max(occ['slots_reject'], 1):5.1f} of 64 lanes")
        print("   iterations by lanes holding a path: none %.1f %%, 1-16 %.1f %%, 17-48 %.1f %%, 49-64 %.1f %%" % tuple(100.0 * c[k] / iters for k in ("Sd", "N", "P", "rand_calls")))
        rt.close()
        return
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
        run("C3", 3840, 2160, 64, bench_scene("C3"), R.RT_SHADE_CHECKER, math)
        run("C4", 1920, 1080, 4, bench_scene("C4"), 0, math)
        run("C2", 1920, 1080, 16, scenes.default_scene(1), 0, math)
        run("4K default scene", 3840, 2160, 64, scenes.default_scene(1), 0, math)
