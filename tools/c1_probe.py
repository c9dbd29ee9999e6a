"""C1 (640x360, 1 spp x 64 frames: the reference's own window at its own sample count) as one 64-frame render, timed a few times;
under rocprofv3 --kernel-trace the kernel durations say where its 2 ms go. Scratch probe."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import raytracer_amd as R
from raytracer_amd import configs, host

cfg = configs.get("C1")
W, H, spp, frames = int(os.environ.get("PROBE_W", cfg["width"])), int(os.environ.get("PROBE_H", cfg["height"])), int(os.environ.get("PROBE_SPP", cfg["spp"])), cfg["frames"]
tb = host.time_table(frames)
rt = R.Raytracer(W, H, cfg["flags"]); rt.set_scene(cfg["spheres"], cfg["materials"])
cam = host.camera_raw(W, H, spp=spp, frames=frames)
for rep in range(3):
    rt.clear(); cam["current_frame"] = 0; rt.set_camera(cam); rt.sync()
    t0 = time.perf_counter()
    rt.render_frames(tb)
    t1 = time.perf_counter()
    rt.sync()
    t2 = time.perf_counter()
    print(f"render {rep}: call {1e3 * (t1 - t0):.3f} ms, sync {1e3 * (t2 - t1):.3f} ms, total {1e3 * (t2 - t0):.3f} ms", flush=True)
rt.close()

# back to back, one sync at the end: does the 1.7 ms kernel shrink when the GPU is kept busy (clock ramp)?
rt = R.Raytracer(W, H, cfg["flags"]); rt.set_scene(cfg["spheres"], cfg["materials"])
rt.set_camera(cam); rt.render_frames(tb); rt.sync()
rt.enable_timing(True); rt.timing_reset()
t0 = time.perf_counter()
for rep in range(int(os.environ.get('PROBE_REPS', '200'))):
    cam["current_frame"] = 0; rt.set_camera(cam)
    rt.render_frames(tb)
rt.sync()
dt = time.perf_counter() - t0
ms, n = rt.timing()
print(f"{W}x{H} {spp} spp x {frames} frames, renders back to back: {1e3 * dt / int(os.environ.get('PROBE_REPS', '200')):.3f} ms per render (wall), path kernel {ms / max(n, 1):.3f} ms per launch over {n} launches", flush=True)
rt.close()
