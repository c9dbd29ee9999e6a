#!/usr/bin/env python3
"""Runs the five BASELINE configs end to end on one GPU (whole render, batched launches of <= 64
frames) in strict and fast math; prints a markdown table for BASELINE.md."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import raytracer_amd as R
from raytracer_amd import configs, host

def flops(c):
    return (17*c["T"] + 4*c["D"] + 12*c["H"] + 6*c["R"] + 15*c["M"] + 8*c["S"] + 12*c["Sm"] + 11*c["J"] + 7*c["Sd"] + 8*c["N"] + 110*c["P"])

def run(name, math, frames_override=None, chunk=64):
    cfg = configs.get(name)
    W, H, spp = cfg["width"], cfg["height"], cfg["spp"]
    frames = frames_override or cfg["frames"]
    flags = cfg["flags"] | (R.RT_MATH_FAST if math == "fast" else 0)
    tb = host.time_table(frames)
    # counters from one instrumented frame
    rt = R.Raytracer(W, H, flags); rt.set_scene(cfg["spheres"], cfg["materials"])
    cam = host.camera_raw(W, H, spp=spp, frames=frames)
    rt.enable_counters(True); rt.set_camera(cam); rt.set_time(int(tb[0])); rt.compute(); rt.sync()
    c = rt.counters(); rt.enable_counters(False); rt.clear()
    rt.set_camera(cam); rt.render_frames(tb[:1]); rt.sync(); rt.clear()   # warm
    cam["current_frame"] = 0; rt.set_camera(cam)
    t0 = time.perf_counter()
    for s in range(0, frames, chunk):
        rt.render_frames(tb[s:s + chunk])
    rt.sync()
    dt = time.perf_counter() - t0
    rt.close()
    samples = c["N"] * frames
    print(f"| {name} {W}x{H} {spp} spp x {frames} fr, {cfg['spheres'].shape[0]} spheres | 1 | {math} | {dt*1e3:9.1f} ms | {samples/dt/1e9:7.2f} G | {c['R']*frames/dt/1e9:7.2f} G | "
          f"{flops(c)*frames/dt/1e12:6.2f} | {flops(c)*frames/dt/1e12/157.3*100:5.1f} % | {flops(c)*frames/dt/1e12/78.6*100:5.1f} % |", flush=True)

if __name__ == "__main__":
    print("| config | GPUs | math | wall | samples/s | rays/s | algorithmic TFLOP/s | % of 157.3 | % of 78.6 |")
    print("|---|---|---|---|---|---|---|---|---|")
    for math in ("strict", "fast"):
        run("C1", math)
        run("C2", math)
        run("C3", math)
        run("C4", math)
        run("C5", math)   # all 1024 frames on ONE GPU (the 8-GPU frame split of BASELINE config 5 is bench.py --config C5 --gpus 8)
