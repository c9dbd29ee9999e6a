#!/usr/bin/env python3
"""A stand-in for the 1 -> 8 GPU curve on the ONE GPU there is: every rank's partition of the bench
workloads timed on one MI355X, one after the other, with the bench's launch shape.

  tiles  (BASELINE config 3): rank r of N renders the 8x8 tiles t = r (mod N) of every C3 frame; a render of K passes is K
         recorded rt_compute passes run as batched launches. T_r(N) = wall time of rank r's K passes (synchronised).
         predicted efficiency = T(1) / (N x max_r T_r(N)); + pack / unpack of the tile gather and a 1-rank RCCL gather of
         one rank's real payload.
  frames (BASELINE config 5): rank r of N renders frames f = r (mod N) of the whole 8K image, 1 spp, into a float sum:
         per-GPU work does not depend on N (weak scaling); predicted efficiency = T(1) / max_r T_r(N); + the reduce's payload.

PREDICTED, NOT MEASURED ON 8 GPUS: ranks run alone on the chip here (no shared-host effects, no xGMI), and the collective
is timed with one rank only. Writes markdown to stdout.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (first, as in bench.py: torch must find the GPU before the library initialises HIP)

torch.cuda.init()

import raytracer_amd as R  # noqa: E402
from raytracer_amd import configs, host  # noqa: E402


def time_rank(cfg, flags, rank, world, split, passes, reps):
    """wall seconds of `passes` recorded passes of rank `rank` of `world`, best and mean of `reps` (after one warm-up render)"""
    W, H, spp = cfg["width"], cfg["height"], cfg["spp"]
    rt = R.Raytracer(W, H, flags)
    rt.set_scene(cfg["spheres"], cfg["materials"])
    if split == "tiles":
        rt.set_partition(rank, world)
    n = passes * (reps + 1)
    tb = host.time_table(n * (world if split == "frames" else 1))
    cam = host.camera_raw(W, H, spp=spp, frames=n * world)
    times, at = [], 0
    for rep in range(reps + 1):
        rt.sync()
        t0 = time.perf_counter()
        for i in range(at, at + passes):
            f = i * world + rank if split == "frames" else i
            rt.set_time(int(tb[f]))
            cam["current_frame"] = f
            rt.set_camera(cam)
            rt.compute()
        rt.sync()
        if rep:
            times.append(time.perf_counter() - t0)
        at += passes
    rt.close()
    return min(times), sum(times) / len(times)


def gather_costs(W, H, world):
    """pack (this rank's own tiles -> send buffer), unpack (world - 1 received buffers -> image) and a 1-rank RCCL gather of
    one rank's payload, on the device; milliseconds"""
    import torch
    import torch.distributed as dist

    from raytracer_amd import dist as rdist

    dev = "cuda:0"
    img = torch.rand((H, W, 4), dtype=torch.float32, device=dev)
    g = rdist.TileGather(img, rank=world - 1, world=world, dst=world - 1)  # as the destination rank: it packs AND unpacks
    def timed(fn, n=20):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    nmine = g.mine[0].numel()
    def pack():
        g.send[:nmine] = g.tiles[g.mine[0], g.mine[1]]
    def unpack():
        for r in range(world - 1):
            ty, tx = g.theirs[r]
            g.tiles[ty, tx] = g.recv[r][: ty.numel()]
    t_pack, t_unpack = timed(pack), timed(unpack)
    payload = g.send.numel() * g.send.element_size()
    t_coll = None
    try:
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
        send = torch.zeros_like(g.send)
        recv = [torch.zeros_like(g.send)]
        t_coll = timed(lambda: dist.gather(send, recv, dst=0))
    except Exception as e:  # noqa: BLE001
        print(f"(RCCL gather with one rank unavailable: {e})", file=sys.stderr)
    return t_pack, t_unpack, t_coll, payload


def reduce_cost(W, H):
    import torch
    import torch.distributed as dist

    t = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3, t.numel() * 4


def main():
    passes = int(os.environ.get("PROXY_PASSES", "20"))
    reps = int(os.environ.get("PROXY_REPS", "3"))
    print("# Scaling proxy on ONE MI355X - predicted, not measured on 8 GPUs\n")
    print("Every rank's partition is rendered alone on the one GPU this round has, with the bench's launch shape (recorded passes, batched")
    print(f"launches). T = wall time of {passes} passes, best of {reps} repetitions after a warm-up render. The driver's 8-GPU node adds what one GPU")
    print("cannot show: ranks sharing a host, the collective over xGMI, clock differences between devices (MI355X_MICROARCH.md: up to 12 %).\n")

    cfg = configs.get("C3")
    flags = cfg["flags"] & ~R.RT_COMBINE_SUM
    print(f"## tiles split, C3 (3840x2160, 64 spp, 20 spheres, strict), {passes}-pass renders - strong scaling\n")
    print("| N | tiles per rank | T_r(N) per rank, ms (best) | max | mean | N x max / T(1) | predicted efficiency T(1) / (N x max) |")
    print("|---|---|---|---|---|---|---|")
    t1 = None
    for world in (1, 2, 4, 8):
        ts = [time_rank(cfg, flags, r, world, "tiles", passes, reps)[0] * 1e3 for r in range(world)]
        if world == 1:
            t1 = ts[0]
        tiles = (3840 // 8) * (2160 // 8) // world
        print(f"| {world} | {tiles} | " + " ".join(f"{t:.1f}" for t in ts) + f" | {max(ts):.1f} | {sum(ts) / len(ts):.1f} | {world * max(ts) / t1:.3f} | {t1 / (world * max(ts)):.3f} |", flush=True)
    print("\nWhat longer batches buy at N = 8 (the staging buffer is indexed by the rank's own tiles since round 3: 16.6 MB per pass at 4K / 8,")
    print("so the batch limit is the library's 64 recorded passes instead of the 32 a whole-image staging buffer allowed):\n")
    print("| passes per render (= per launch up to 64) | rank 3 of 8: ms per pass | rank 0 of 1: ms per pass |")
    print("|---|---|---|")
    for k in (1, 8, 20, 32, 64):
        a = time_rank(cfg, flags, 3, 8, "tiles", k, reps)[0] * 1e3 / k
        b = time_rank(cfg, flags, 0, 1, "tiles", k, reps)[0] * 1e3 / k if k <= 32 else float("nan")
        print(f"| {k} | {a:.3f} | {b:.3f} |", flush=True)
    try:
        tp, tu, tc, payload = gather_costs(3840, 2160, 8)
        print(f"\nTile gather at N = 8, float image: payload {payload / 1e6:.1f} MB per rank; pack {tp:.3f} ms, unpack of 7 ranks' tiles {tu:.3f} ms"
              + (f", 1-rank RCCL gather of one payload {tc:.3f} ms" if tc is not None else "") +
              f"; over xGMI every rank sends its {payload / 1e6:.1f} MB on its own link to rank 0 at once (~153 GB/s per link: {payload / 153e9 * 1e3:.2f} ms).")
    except Exception as e:  # noqa: BLE001
        print(f"\n(tile gather timing failed: {e})")

    cfg = configs.get("C5")
    flags = cfg["flags"] | R.RT_ACCUM_F32 | R.RT_COMBINE_SUM
    kf = int(os.environ.get("PROXY_C5_PASSES", "64"))
    print(f"\n## frames split, C5 (7680x4320, 1 spp, default scene, float sum), {kf} frames per rank - weak scaling\n")
    print("| N | frame set | T_r(N), ms per rank (best) | max | predicted efficiency T(1) / max |")
    print("|---|---|---|---|---|")
    t1 = None
    for world in (1, 8):
        ranks = range(world)
        ts = [time_rank(cfg, flags, r, world, "frames", kf, 2)[0] * 1e3 for r in ranks]
        if world == 1:
            t1 = ts[0]
        print(f"| {world} | f = r (mod {world}) | " + " ".join(f"{t:.1f}" for t in ts) + f" | {max(ts):.1f} | {t1 / max(ts):.3f} |", flush=True)
    try:
        tr, nbytes = reduce_cost(7680, 4320)
        print(f"\nReduce payload: {nbytes / 1e6:.1f} MB per GPU, once per render; a 1-rank RCCL reduce of it takes {tr:.3f} ms (no data moves: the call's fixed cost). "
              f"Ring over xGMI: ~(7/8) x {nbytes / 1e6:.0f} MB / 153 GB/s = {0.875 * nbytes / 153e9 * 1e3:.1f} ms, against {t1:.0f} ms of tracing per {kf} frames and rank.")
    except Exception as e:  # noqa: BLE001
        print(f"\n(reduce timing failed: {e})")


if __name__ == "__main__":
    main()
