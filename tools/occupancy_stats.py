import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import raytracer_amd as R, oracle_lib as O, scenes
sp, mt = scenes.random_bunch(20, 3, 3)
rt = R.Raytracer(3840, 2160, R.RT_ACCUM_F32 | R.RT_SHADE_CHECKER)
rt.set_scene(sp, mt); cam = scenes.camera_raw(3840, 2160, spp=64, frames=10)
rt.enable_counters(True); rt.set_camera(cam); rt.set_time(int(O.time_bits_table(1)[0])); rt.compute(); rt.sync()
c = rt.counters(); o = rt.occupancy()
wt, wr = o["slots_trace"] / 64, o["slots_reject"] / 64
print(c)
print(f"trace wave-iterations {wt:.0f}, rejection rounds executed {wr:.0f} = {wr/wt:.2f} per trace iteration; tries J {c['J']} = {c['J']/wr:.1f} lanes per round; "
      f"diffuse hits Sd {c['Sd']} = {c['Sd']/wt:.1f} per iteration; new rays N {c['N']} = {c['N']/wt:.1f} per iteration; rays {c['R']/wt:.1f} per iteration")
