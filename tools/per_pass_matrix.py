"""One launch per pass (rt_compute + rt_sync per frame: a host that presents every frame) over image sizes and samples per pixel, default
scene: wall per pass, path kernel per launch and samples/s - a scan for launch shapes that fall off the throughput the batched bench shows."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import raytracer_amd as R
from raytracer_amd import configs, host

cfg = configs.get("C1")
print("| image | spp | wall per pass | path kernel | samples/s |")
print("|---|---|---|---|---|")
for W, H in ((640, 360), (1280, 720), (1920, 1080), (3840, 2160), (7680, 4320)):
    for spp in (1, 4, 16, 64):
        if W * H * spp > 7680 * 4320 * 16:
            continue
        frames = 24
        tb = host.time_table(frames)
        rt = R.Raytracer(W, H, R.RT_ACCUM_F32); rt.set_scene(cfg["spheres"], cfg["materials"])
        cam = host.camera_raw(W, H, spp=spp, frames=frames)
        rt.enable_timing(True)
        for f in range(frames):
            if f == 8:
                rt.sync(); rt.timing_reset(); t0 = time.perf_counter()
            cam["current_frame"] = f
            rt.set_time(int(tb[f])); rt.set_camera(cam); rt.compute(); rt.sync()
        dt = (time.perf_counter() - t0) / (frames - 8)
        ms, n = rt.timing()
        cov = (W // 16) * (H // 16) * 256
        print(f"| {W}x{H} | {spp} | {1e3 * dt:.3f} ms | {ms / max(n, 1):.3f} ms | {cov * spp / dt / 1e9:.2f} G |", flush=True)
        rt.close()
