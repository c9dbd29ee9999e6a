#!/usr/bin/env python3
"""What a render costs beyond its passes: wall time of K recorded C3 passes (one batched launch) for rank r of N, K = 1, 8, 20, 64;
T(K) = K t + F. usage: tools/r3_fixed_cost.py [rank world ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from scaling_proxy import time_rank
import raytracer_amd as R
from raytracer_amd import configs

cfg = configs.get(os.environ.get("FC_CONFIG", "C3"))
flags = cfg["flags"]
parts = [(0, 1), (3, 8)] if len(sys.argv) < 3 else [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
for rank, world in parts:
    row = []
    for k in (1, 8, 20, 64):
        if world == 1 and k == 64 and cfg["width"] >= 3840:
            continue
        best, mean = time_rank(cfg, flags, rank, world, "tiles", k, 4)
        row.append((k, best * 1e3))
    (k0, t0), (k1, t1) = row[-2], row[-1]
    t = (t1 - t0) / (k1 - k0); F = t1 - k1 * t
    print(f"rank {rank} of {world}: " + ", ".join(f"K={k}: {ms:.2f} ms ({ms / k:.3f} per pass)" for k, ms in row) + f"  ->  t = {t:.3f} ms per pass, F = {F:.2f} ms per render", flush=True)
