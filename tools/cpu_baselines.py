#!/usr/bin/env python3
"""CPU baseline plan of BASELINE.md: the oracle (scalar C, OpenMP) on the host cores."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_lib as O
from raytracer_amd import configs, host

threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
print(f"threads: {threads}")
print("| config | cores | wall | samples/s | rays/s | samples/s/core | T D H R M S Sm J Sd N P |")
print("|---|---|---|---|---|---|---|")
for name, spp, frames in (("C1", 1, 64), ("C2", 16, 8), ("C3", 64, 1), ("C4", 1, 2)):
    cfg = configs.get(name)
    W, H = cfg["width"], cfg["height"]
    oflags = cfg["flags"] & (O.ACCUM_F32 | O.DISPATCH_FULL | O.SHADE_CHECKER)
    img = O.OracleImage(W, H, oflags)
    cam = host.camera_raw(W, H, spp=spp, frames=frames)
    t0 = time.perf_counter()
    O.render(img, cfg["spheres"], cfg["materials"], cam, 0, frames, threads=threads)
    dt = time.perf_counter() - t0
    c = img.counters_dict()
    print(f"| {name} {W}x{H} {spp} spp x {frames} fr | {threads} | {dt:.2f} s | {c['N']/dt/1e6:.1f} M | {c['R']/dt/1e6:.1f} M | {c['N']/dt/threads/1e6:.2f} M | "
          + " ".join(str(c[k]) for k in ("T", "D", "H", "R", "M", "S", "Sm", "J", "Sd", "N", "P")) + " |", flush=True)
