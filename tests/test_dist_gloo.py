"""N > 1 path on CPU: world_size-2 gloo processes exercise the partition rules and the combine
collective of raytracer_amd/dist.py. The per-rank images come from the oracle's tile/frame
partition (the GPU kernel cannot run here); the GPU tests check that the HIP partition produces
exactly those images."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
import scenes

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracer_amd import dist as rdist

    W, H, spp, frames = 80, 64, 2, (6 if world < 8 else 16)
    spheres, mats = scenes.default_scene(1)
    cam = scenes.camera_raw(W, H, spp=spp, frames=frames)
    try:
        gather = mode.endswith("_gather")
        if gather:
            mode = mode[: -len("_gather")]
        if mode in ("tiles_f32", "tiles_rgba8"):
            flags = O.ACCUM_F32 if mode == "tiles_f32" else 0
            img = O.OracleImage(W, H, flags)
            O.render(img, spheres, mats, cam, 0, frames, rank=rank, world=world, threads=1)
            mask = rdist.tile_owner_mask(W, H, world, rank)
            mine = img.accum if flags else img.rgba8
            assert not mine[~mask].any() and mine[mask][..., 3].all()  # my tiles rendered, others zero
            t = torch.from_numpy(mine.copy())
            if gather:  # the same combine as a gather of each rank's own tiles
                assert rdist.TileGather.usable(W, H)
                rdist.TileGather(t, rank, world, dst=0).run()
            else:
                rdist.combine_sum(t, dst=0)
            if rank == 0:
                whole = O.OracleImage(W, H, flags)
                O.render(whole, spheres, mats, cam, 0, frames, threads=1)
                want = whole.accum if flags else whole.rgba8
                q.put(bool(np.array_equal(t.numpy().view(np.uint8), want.view(np.uint8))))
        elif mode == "frames":
            flags = O.ACCUM_F32 | O.COMBINE_SUM
            tb = O.time_bits_table(frames)
            mine_frames = rdist.frame_set(frames, world, rank)
            img = O.OracleImage(W, H, flags)
            for f in mine_frames:
                c = np.array(cam, copy=True); c["current_frame"] = f
                O.compute(img, spheres, mats, c, int(tb[f]), threads=1)
            t = torch.from_numpy(img.accum.copy())
            rdist.combine_sum(t, dst=0)
            if rank == 0:
                avg = rdist.finish_frame_split(t, frames).numpy()
                whole = O.OracleImage(W, H, O.ACCUM_F32)
                O.render(whole, spheres, mats, cam, 0, frames, threads=1)
                ok = np.abs(avg - whole.accum).max() < 2e-6 and np.all(t.numpy()[..., 3] == frames)
                q.put(bool(ok))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("tiles_f32", 2), ("tiles_rgba8", 2), ("frames", 2), ("tiles_f32_gather", 2),
                                        ("tiles_rgba8_gather", 2), ("tiles_f32_gather", 3), ("tiles_f32_gather", 8), ("frames", 8)])
def test_two_rank_combine(mode, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_partition_rules_cover_everything_once():
    from raytracer_amd import dist as rdist

    for (W, H, world) in [(80, 64, 2), (70, 50, 3), (3840, 2160, 8)]:
        total = np.zeros((H, W), np.int32)
        for r in range(world):
            total += rdist.tile_owner_mask(W, H, world, r)
        cov = np.zeros((H, W), np.int32)
        cov[: H // 16 * 16, : W // 16 * 16] = 1
        assert np.array_equal(total, cov)
        sizes = [rdist.tile_owner_mask(W, H, world, r).sum() for r in range(world)]
        assert max(sizes) - min(sizes) <= 64 * (1 + (W // 8) // world)  # interleaved tiles balance the ranks
    assert sorted(np.concatenate([rdist.frame_set(10, 3, r) for r in range(3)]).tolist()) == list(range(10))
