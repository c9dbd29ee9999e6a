#!/usr/bin/env python3
"""bench.py — the contract benchmark: samples/s of the path-tracing compute pass on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A STEP is one compute pass (`rt_compute`, reference Raytracer::compute, src/raytracing.rs:178-200)
over the whole image of the workload BASELINE.json's metric is quoted on: C3 = 3840x2160, 64 spp,
glowing-checkers scene (random_bunch(20)). rt_compute, like the reference's compute, records the pass;
the library runs up to 64 recorded passes (32 at 4K) as one batched launch when the caller synchronises (bit-identical
to one launch per pass; `per_pass_launches` in the JSON line is the same measurement with RT_DEFER_FRAMES=1).
The K timed steps are one render; the render is repeated (each repetition bracketed and timed on its own) until at
least three seconds have been measured, and `value` is K steps' samples over the MEAN repetition time (`repeats`).
With N GPUs the default partition is BASELINE config 3's: image tiles (`--split tiles`: rank r renders the 8x8 tiles
t = r mod N of every frame, "strong" scaling on the fixed 4K frame, bit-identical to the 1-GPU image), the per-GPU
images combined by ONE RCCL collective at the end of the render, inside the timed region (a gather of each rank's own
tiles onto rank 0; RT_BENCH_COMBINE=reduce uses a sum-reduce of the whole images instead - the same bits). `--split frames` (the
default for C5, BASELINE config 5) splits the progressive render by disjoint per-frame sample sets instead: in step i
rank r renders frame i*N + r of the whole image into a float sum, per-GPU work is fixed ("weak" scaling) and a step
of the job is N frames; the N > 1 line carries the other split's measurement as `other_split`.

Prints ONE JSON line (rank 0). `value` = covered pixels x spp x K / seconds, whole job, inputs
resident in HBM, RT_MATH_STRICT (the mode that is bit-identical to the oracle). Extra objects:
  roofline     FP32 VALU roofline of the trace kernel, against 157.3 TFLOP/s. `frac` = the definition: FLOPs of the
               REFERENCE ALGORITHM per launch (SURVEY.md §8d formula over the device event counters of a counting
               launch that traces every tile) / average launch duration (hipEvents on the launch stream).
               `frac_executed` = the same over the events the production launch really executes (rt_enable_counters
               mode 2). `valu_busy` / `traffic` are NOT measured by this run: they are read from the committed
               rocprofv3 --pmc passes of this command (profiles/valu_busy_*.json, traffic_*.json; `*_source` says so).
  cpu_baseline the oracle (scalar C port, OpenMP) timed on this box's host cores on a bounded
               tile sample of the same frame (rank 0, N = 1 only).
  fast_math    the same measurement in RT_MATH_FAST (image within tolerance, not bit-exact).
  per_pass_launches the same measurement with one kernel launch per pass (RT_DEFER_FRAMES=1; N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md:41: 256 CU x 256 FLOP/clk x 2.4 GHz


def algorithmic_flops(c):
    """17T + 4D + 12H + 6R + 15M + 8S + 12Sm + 11J + 7Sd + 8N + 110P."""
    return (17 * c["T"] + 4 * c["D"] + 12 * c["H"] + 6 * c["R"] + 15 * c["M"] + 8 * c["S"] + 12 * c["Sm"]
            + 11 * c["J"] + 7 * c["Sd"] + 8 * c["N"] + 110 * c["P"])


def executed_flops(c):
    """FLOPs of the events the production launch executes (rt_enable_counters mode 2), weighted like SURVEY.md §8d: a phase-1
    filter test is 8 FMAs = 16, an exact hit_sphere test up to its discriminant 17 (the §8d figure), the rest as in §8d."""
    return (16 * c["T"] + 17 * c["X"] + 4 * c["D"] + 12 * c["H"] + 6 * c["R"] + 15 * c["M"] + 8 * c["S"] + 12 * c["Sm"]
            + 11 * c["J"] + 7 * c["Sd"] + 8 * c["N"] + 110 * c["P"])


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=8)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--config", default="C3", help="C2 | C3 | C4 (C3 is the headline: 4K, 64 spp)")
    p.add_argument("--math", default="strict", choices=["strict", "fast"])
    p.add_argument("--split", default=None, choices=["tiles", "frames"],
                   help="N > 1: tiles = interleaved 8x8 tiles of every frame (strong scaling; default, BASELINE config 3); "
                        "frames = disjoint frame sets of the whole image per GPU (weak scaling; default for C5)")
    p.add_argument("--min-seconds", type=float, default=3.0, help="repeat the K-step render until this much time is measured")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="skip the fast_math side measurement")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the oracle sample")
    return p.parse_args()


def measure(R, cfg, flags, rank, world, steps, warmup, split, dist, torch, device, use_dist, min_seconds=1.0, max_repeats=64):
    """Times `steps` compute passes, repeated until `min_seconds` are measured. Returns dict(seconds = mean time of
    one repetition, repeats, kernel_ms, launches, flops_per_launch, ...)."""
    from raytracer_amd import dist as rdist
    from raytracer_amd import host

    W, H, spp = cfg["width"], cfg["height"], cfg["spp"]
    frames_total = warmup + steps * max_repeats + 4
    rt = R.Raytracer(W, H, flags, device=device)
    rt.set_scene(cfg["spheres"], cfg["materials"])
    img = rdist.DeviceImage(rt, f"cuda:{device}") if use_dist else None
    if split == "tiles":
        rt.set_partition(rank, world)
    # tile split: rank 0 needs only each rank's OWN tiles - a gather of 1/N of an image per rank instead of a sum-reduce of
    # N whole images (same bits; RT_BENCH_COMBINE=reduce keeps the reduce)
    gather = None
    if img is not None and split == "tiles" and os.environ.get("RT_BENCH_COMBINE", "gather") == "gather" \
            and rdist.TileGather.usable(W, H, bool(flags & R.RT_DISPATCH_FULL)):
        gather = rdist.TileGather(img.tensor, rank, world, dst=0)
    tb = host.time_table(frames_total * (world if split == "frames" else 1))
    cam = host.camera_raw(W, H, spp=spp, frames=frames_total * world)

    def step(i):
        # App::update + App::render for redraw i (reference app.rs:93-116)
        f = i * world + rank if split == "frames" else i
        rt.set_time(int(tb[f]))
        cam["current_frame"] = f
        rt.set_camera(cam)
        rt.compute()

    # algorithmic FLOPs per launch: instrumented variant on the first warmup-like frame (untimed)
    rt.enable_counters(True)
    step(0)
    rt.sync()
    counters = rt.counters()
    # what the production launch EXECUTES (rt_enable_counters mode 2: same loop shape, tile masks / lists, drops and culls as the
    # timed launches, with counters): phase-1 tests run, exact tests run, rays traced, samples finished in sky tiles
    executed = None
    try:
        rt.enable_counters(2)
        rt.reset_counters()
        step(0)
        rt.sync()
        executed = dict(rt.counters())
        occ = rt.occupancy()
        executed["X"], executed["sky_samples"] = occ["slots_trace"], occ["slots_reject"]
    except R.RtError as e:  # scenes without a SPLIT shape (33-64 spheres) have no such variant
        print(f"bench.py: executed-work counters unavailable ({e})", file=sys.stderr)
        rt.sync()
    rt.enable_counters(False)
    rt.reset_counters()
    rt.clear()
    if img is not None:
        img.zero_()
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i)
    rt.sync()
    if use_dist:  # the collective's first call sets up RCCL's channels: part of the warm-up, like the kernels' first launches
        if gather is not None:
            try:
                gather.run()
            except (RuntimeError, ValueError, NotImplementedError) as e:  # a backend without gather: every rank lands here alike
                print(f"bench.py: tile gather unavailable ({e}); using the sum-reduce", file=sys.stderr)
                gather = None
        if gather is None:
            rdist.combine_sum(img.tensor, dst=0)
        torch.cuda.synchronize()
    rt.enable_timing(True)
    rt.timing_reset()
    total, repeats, at = 0.0, 0, warmup
    while True:
        # one repetition = EXACTLY `steps` steps, bracketed by barrier + synchronize on both sides
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(at, at + steps):
            step(i)
        rt.sync()
        if gather is not None:
            gather.run()                          # ONE RCCL gather per render (tile split)
        elif use_dist:
            rdist.combine_sum(img.tensor, dst=0)  # ONE RCCL reduce per render
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        total += time.perf_counter() - t0
        repeats += 1
        at += steps
        go_on = torch.tensor([1.0 if (total < min_seconds and repeats < max_repeats) else 0.0], device=f"cuda:{device}")
        if use_dist and world > 1:
            dist.all_reduce(go_on, op=dist.ReduceOp.MAX)  # every rank runs the same number of repetitions (outside the timed region)
        if go_on.item() == 0.0:
            break
    seconds = total / repeats
    kernel_ms, launches = rt.timing()
    sky_tiles, my_tiles = rt.sky_tile_stats()
    info = rt.device_info()
    if img is not None:
        rt.bind_accum(0, 0)
    rt.close()
    cov_w = W if flags & R.RT_DISPATCH_FULL else (W // 16) * 16
    cov_h = H if flags & R.RT_DISPATCH_FULL else (H // 16) * 16
    samples_per_step = cov_w * cov_h * spp * (world if split == "frames" else 1)
    return {"combine": "gather" if gather is not None else ("reduce" if use_dist else None),
            "seconds": seconds, "repeats": repeats, "kernel_ms": kernel_ms, "launches": launches, "counters": counters,
            "flops_per_launch": algorithmic_flops(counters), "samples_per_step": samples_per_step, "device": info,
            "sky_tiles": sky_tiles, "my_tiles": my_tiles,
            "executed": executed, "executed_flops_per_launch": executed_flops(executed) if executed else None}


def cpu_baseline(cfg, flags, target_seconds):
    """The oracle on this box's host cores, on every `world`-th 8x8 tile of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from raytracer_amd import host

    W, H, spp = cfg["width"], cfg["height"], cfg["spp"]
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    oflags = flags & (O.ACCUM_F32 | O.DISPATCH_FULL | O.SHADE_CHECKER)
    cam = host.camera_raw(W, H, spp=spp, frames=8)
    tb = host.time_table(1)

    def run(world):
        img = O.OracleImage(W, H, oflags)
        t0 = time.perf_counter()
        O.compute(img, cfg["spheres"], cfg["materials"], cam, int(tb[0]), rank=0, world=world, threads=threads)
        return time.perf_counter() - t0, img.counters_dict()

    # calibrate on growing tile samples until one takes >= 0.5 s, then size the timed sample
    total = (W // 16) * 16 * (H // 16) * 16 * spp
    world = 4096
    while True:
        dt, c = run(world)
        if dt >= 0.5 or world == 1:
            break
        world = max(1, world // 8)
    rate = c["N"] / max(dt, 1e-6)
    want = int(max(1, min(4096, round(total / max(rate * target_seconds, 1.0)))))
    frames = 1
    if want != world:
        world = want
        dt, c = run(world)
    if world == 1 and dt < 0.6 * target_seconds:  # a whole frame is too short: time several
        frames = int(min(8, max(2, round(target_seconds / dt))))
        n = r = 0
        t0 = time.perf_counter()
        for _ in range(frames):
            _dt, cc = run(1)
            n += cc["N"]; r += cc["R"]
        dt = time.perf_counter() - t0
        c = {"N": n, "R": r}
    return {"value": c["N"] / dt, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{frames} frame(s), every {world}-th 8x8 tile of the {W}x{H} image at {spp} spp "
                      f"({c['N']} samples, {c['R']} rays, {dt:.1f} s), oracle/rt_oracle.c with OpenMP",
            "rays_per_s": c["R"] / dt}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    # stdout carries exactly ONE line, the JSON: libraries that print banners to fd 1 (RCCL prints its version
    # there when it initialises) are sent to stderr, and the line goes out through a saved copy of the descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    # RT_BENCH_FORCE_DIST=1 runs the N > 1 code path (torch-owned image, RCCL reduce) with one rank
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))

    import raytracer_amd as R
    from raytracer_amd import configs

    cfg = configs.get(args.config)
    if args.split is None:  # BASELINE: config 3 (and the 1080p ones) by image tiles, config 5 by frame sets
        args.split = "frames" if args.config == "C5" else "tiles"

    def flags_for(split):
        f = cfg["flags"]
        if split == "frames" and use_dist:  # (RT_BENCH_FORCE_DIST=1 rehearses the N > 1 path with one rank)
            return f | R.RT_ACCUM_F32 | R.RT_COMBINE_SUM  # per-GPU frame sums, one reduce, x 1/n afterwards
        return f & ~R.RT_COMBINE_SUM

    base_flags = flags_for(args.split)

    def run(math, split=None):
        split = split or args.split
        flags = flags_for(split) | (R.RT_MATH_FAST if math == "fast" else R.RT_MATH_STRICT)
        m = measure(R, cfg, flags, rank, world, args.steps, args.warmup, split, dist, torch, local_rank, use_dist, args.min_seconds)
        sec = torch.tensor([m["seconds"]], dtype=torch.float64, device=f"cuda:{local_rank}")
        if world > 1:
            dist.all_reduce(sec, op=dist.ReduceOp.MAX)
        m["seconds"] = float(sec.item())
        return m

    m = run(args.math)
    extras = per_pass = other_split = None
    if use_dist and not args.no_extras:
        other_split = run(args.math, "frames" if args.split == "tiles" else "tiles")
    if not args.no_extras and world == 1:
        extras = run("fast" if args.math == "strict" else "strict")
        os.environ["RT_DEFER_FRAMES"] = "1"  # read by rt_create
        per_pass = run(args.math)
        del os.environ["RT_DEFER_FRAMES"]

    if rank == 0:
        def committed(kind, math):
            """A figure that this run does NOT measure: read from the committed PMC pass of this same command under profiles/."""
            path = os.path.join("profiles", f"{kind}_{args.config}_{math}.json")
            try:
                return json.load(open(os.path.join(ROOT, path))), path
            except (OSError, ValueError):
                return None, None

        def roof(mm, math=None, brief=False):
            math = math or args.math
            launches = max(mm["launches"], 1)
            fpl = args.steps * mm["repeats"] / launches  # passes (frames) per kernel launch
            avg_ms = mm["kernel_ms"] / launches
            achieved = mm["flops_per_launch"] * fpl / (avg_ms * 1e-3) / 1e12
            px = cfg["width"] // 16 * 16 * (cfg["height"] // 16 * 16)
            f32 = bool(base_flags & R.RT_ACCUM_F32)
            # path kernel: the image read + written once per launch, or - in a staged batch (the library's rule: several
            # passes, >= 4 spp, fewer than 32 tiles per resident wave) - 16 B per pixel and pass written to the staging buffer
            tiles = px // 64
            staged = fpl > 1 and cfg["spp"] >= 4 and tiles < 32 * 24 * mm["device"]["compute_units"]
            alg_traffic = px * (16 * fpl if staged else (32 if f32 else 8))
            frac = achieved / FP32_VALU_PEAK_TFLOPS
            note = ("FP32 vector roofline (no MFMA: no dense contraction, and FP32 MFMA shares the vector pipe, profiles/r3_mfma_coexec.txt); "
                    "peak 157.3 TFLOP/s = 64 FLOP per clock and SIMD, reached by v_pk_fma_f32 or by two dual-issued v_fma_f32 alike "
                    "(profiles/r3_valu_ceiling.txt). `frac` is the DEFINITION (SURVEY 8d): FLOPs of the REFERENCE ALGORITHM (every ray "
                    "tests every sphere; device event counters of a counting launch" + (" of rank 0's tiles" if world > 1 else "") +
                    ") / the timed kernel's duration. The timed kernel proves part of that work unnecessary and does not run it (sky "
                    "tiles, primary rays tested against their tile's candidates only, own-sphere drop, behind cull - all bit-identical): "
                    "`frac_executed` counts the events it does execute (rt_enable_counters mode 2; a phase-1 filter test = 16 FLOPs, "
                    "an exact test = 17, the rest as 8d) and is the utilisation figure")
            if frac > 1.0:
                note = "numerator includes work proven unnecessary (frac > 1: the timed kernel does not execute the counted work). " + note
            out = {"bound": "valu", "achieved": round(achieved, 3), "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(frac, 4)}
            if mm["executed_flops_per_launch"] is not None:
                ex = mm["executed_flops_per_launch"] * fpl / (avg_ms * 1e-3) / 1e12
                e = mm["executed"]
                out.update({"achieved_executed": round(ex, 3), "frac_executed": round(ex / FP32_VALU_PEAK_TFLOPS, 4),
                            "executed_per_pass": {"phase1_tests": e["T"], "exact_tests": e["X"], "rays": e["R"], "samples": e["N"],
                                                  "sky_tile_samples": e["sky_samples"], "rejection_tries": e["J"],
                                                  "reference_algorithm_tests": mm["counters"]["T"], "reference_algorithm_rays": mm["counters"]["R"]}})
            vb, vb_path = committed("valu_busy", math)
            if vb is not None and world == 1:
                out.update({"valu_busy": vb["valu_busy"], "valu_dual_issue_share": vb["valu2_share"], "lane_utilisation": vb["lane_utilisation"],
                            "valu_busy_source": f"{vb_path}: rocprofv3 --pmc pass of this command, NOT measured in this run; "
                                                "4 x (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) / (SIMDs x cycles), calibrated in profiles/r3_valu_ceiling.txt"})
            tr, tr_path = committed("traffic", math)
            traffic = tr["bytes_per_pass"] * fpl if (tr is not None and world == 1 and staged and "bytes_per_pass" in tr) else None
            out.update({"sky_tiles": mm["sky_tiles"], "tiles": mm["my_tiles"],
                        "traffic": traffic,
                        "traffic_source": (f"{tr_path}: TCC_EA0 request counters of a separate rocprofv3 --pmc pass of this command, bytes per pass x "
                                           "passes per launch; NOT measured in this run") if traffic is not None else None,
                        "traffic_algorithmic": int(alg_traffic),
                        "kernel": "rt_path_kernel", "avg_launch_ms": round(avg_ms, 4), "launches": launches, "passes_per_launch": fpl,
                        "algorithmic_flops_per_launch": int(mm["flops_per_launch"] * fpl),
                        "note": ("numerator includes work proven unnecessary; " if frac > 1.0 else "") + "see roofline.note" if brief else note})
            return out

        total_samples = m["samples_per_step"] * args.steps
        out = {
            "metric": "samples/sec at 4K 64spp (path-tracing compute pass)" if args.config == "C3" else f"samples/sec ({args.config})",
            "value": total_samples / m["seconds"],
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": m["seconds"] * 1e3 / args.steps,
            "repeats": m["repeats"],
            "higher_is_better": True,
            "scaling": "n/a" if world == 1 else ("strong" if args.split == "tiles" else "weak"),
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{cfg['name']}: {cfg['description']}", "width": cfg["width"], "height": cfg["height"],
                       "spp": cfg["spp"], "spheres": int(cfg["spheres"].shape[0]), "math": args.math,
                       "accum": "f32" if base_flags & R.RT_ACCUM_F32 else "rgba8",
                       "step": "one rt_compute pass (one frame) over the whole image"},
            "rays_per_s": m["counters"]["R"] * (world if args.split == "tiles" else 1) * args.steps / m["seconds"]
            if world == 1 else None,
            "roofline": roof(m),
            "device": m["device"],
        }
        if use_dist:
            out["config"]["split"] = args.split
            out["config"]["parallelism"] = (f"{world} GPUs x interleaved 8x8 tiles of every frame, one RCCL {m['combine']} per render" if args.split == "tiles"
                                            else f"{world} GPUs x disjoint frame sets of the whole image, one RCCL reduce per render")
        if other_split is not None:
            osplit = "frames" if args.split == "tiles" else "tiles"
            out["other_split"] = {"split": osplit, "scaling": "strong" if osplit == "tiles" else "weak",
                                  "value": other_split["samples_per_step"] * args.steps / other_split["seconds"], "unit": "samples/s",
                                  "ms_per_step": other_split["seconds"] * 1e3 / args.steps, "repeats": other_split["repeats"]}
        other = "fast" if args.math == "strict" else "strict"
        if extras is not None:
            out[f"{other}_math"] = {"value": extras["samples_per_step"] * args.steps / extras["seconds"], "unit": "samples/s",
                                    "ms_per_step": extras["seconds"] * 1e3 / args.steps, "roofline": roof(extras, other, brief=True),
                                    "parity": ("tolerance (PSNR >= 40 dB, max-abs <= 3/255 on unchanged trajectories, >= 99 % identical RNG states) "
                                               "holds on C1, C2, C3 and C5; NOT met on C4 (39.3 dB, 91.5 % identical states: "
                                               "tests/test_gpu_golden.py pins what it measures there); RNG function bit-exact everywhere")
                                    if other == "fast" else "bit-identical to the oracle"}
        if per_pass is not None:
            out["per_pass_launches"] = {"value": per_pass["samples_per_step"] * args.steps / per_pass["seconds"], "unit": "samples/s",
                                        "ms_per_step": per_pass["seconds"] * 1e3 / args.steps, "roofline": roof(per_pass, brief=True),
                                        "note": "RT_DEFER_FRAMES=1: every rt_compute is its own kernel launch (each with its own tail)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, base_flags, args.cpu_seconds)
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
