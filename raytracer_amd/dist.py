"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference drives a single device; the split is this build's. Two partitions:

  tiles   rank r renders the 8x8 pixel tiles t with t % world == r (rt_set_partition); other
          pixels of its image stay zero, so ONE sum-reduce of the per-GPU images at the end of
          the render assembles the picture. Bit-identical to the 1-GPU image in every mode.
  frames  rank r renders frame set {f : f % world == r} of the WHOLE image into a float sum
          (RT_COMBINE_SUM); one sum-reduce, then * 1/n_frames. Equals the un-quantised running
          average up to f32 summation order.

No collective runs inside the per-frame path: pixels and frames are independent given
(u, v, time bits). The only exchange is the final reduce, issued once per render.
"""
import numpy as np
import torch
import torch.distributed as dist


def tile_owner_mask(width, height, world, rank, full_dispatch=False):
    """Boolean H x W mask of the pixels rank `rank` renders under the tile split (host-side restatement of
    the kernel's tile -> rank rule; used by tests and by callers that want to gather instead of reduce)."""
    cov_w = width if full_dispatch else (width // 16) * 16
    cov_h = height if full_dispatch else (height // 16) * 16
    tiles_x = (cov_w + 7) // 8
    ys, xs = np.mgrid[0:height, 0:width]
    tile = (ys // 8) * tiles_x + xs // 8
    return (tile % world == rank) & (xs < cov_w) & (ys < cov_h)


def frame_set(n_frames, world, rank):
    """Frame indices rank `rank` renders under the frame-set split."""
    return np.arange(rank, n_frames, world)


def combine_sum(image, dst=0, group=None):
    """Sum-reduce the per-GPU accumulation images onto rank `dst` (RCCL reduce over xGMI, or gloo on
    CPU). `image` is a torch tensor: float32 H x W x 4, or uint8 H x W x 4 for RGBA8 tile images, which
    are reduced as int32 words (tiles are disjoint, so a sum is a merge)."""
    if not dist.is_initialized():
        return image
    buf = image.view(torch.int32) if image.dtype == torch.uint8 else image
    dist.reduce(buf, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return image


class TileGather:
    """The tile split's combine as a GATHER instead of a sum-reduce: rank r owns the 8x8 tiles t with t % world == r
    and nothing else, so rank `dst` only needs each rank's own tiles - 1/world of an image per rank, all ranks sending at
    once over their own xGMI links - instead of `world` whole images through a ring (4K float image: 16.6 MB per rank
    at 8 GPUs against 132.7 MB). The result is the same bits as `combine_sum` (a sum of disjoint tiles is a merge).

    Index tensors and buffers are made once, outside any timed region. Works on the image itself (the kernel's bound
    accumulation image): float32 or uint8 H x W x 4 with H, W multiples of 8 and the whole image covered by the
    dispatch (width and height multiples of 16, or RT_DISPATCH_FULL); `usable()` says whether that holds -
    otherwise use `combine_sum`."""

    @staticmethod
    def usable(width, height, full_dispatch=False):
        return width % 8 == 0 and height % 8 == 0 and (full_dispatch or (width % 16 == 0 and height % 16 == 0))

    def __init__(self, image, rank, world, dst=0, group=None):
        H, W, C = image.shape
        if not (H % 8 == 0 and W % 8 == 0):
            raise ValueError("TileGather needs an image of whole 8x8 tiles")
        self.image, self.rank, self.world, self.dst, self.group = image, rank, world, dst, group
        self.tiles = image.view(H // 8, 8, W // 8, 8, C).permute(0, 2, 1, 3, 4)  # [ty, tx, 8, 8, C], a view
        tiles_x, n_tiles = W // 8, (H // 8) * (W // 8)
        self.per_rank = (n_tiles + world - 1) // world  # ranks with one tile fewer pad their buffer
        dev = image.device

        def index(r):
            t = torch.arange(r, n_tiles, world, device=dev)
            return torch.div(t, tiles_x, rounding_mode="floor"), t % tiles_x

        self.mine = index(rank)
        self.send = torch.zeros((self.per_rank, 8, 8, C), dtype=image.dtype, device=dev)
        self.recv = [torch.zeros_like(self.send) for _ in range(world)] if rank == dst else None
        self.theirs = [index(r) for r in range(world)] if rank == dst else None

    def run(self):
        """Collective: every rank of the group calls it once per render; afterwards `image` on rank `dst` holds the
        whole picture (the other ranks' images are unchanged)."""
        n = self.mine[0].numel()
        self.send[:n] = self.tiles[self.mine[0], self.mine[1]]
        if not dist.is_initialized():
            return self.image
        dist.gather(self.send, self.recv, dst=self.dst, group=self.group)
        if self.rank == self.dst:
            for r in range(self.world):
                if r != self.rank:
                    ty, tx = self.theirs[r]
                    self.tiles[ty, tx] = self.recv[r][: ty.numel()]
        return self.image


def finish_frame_split(image_sum, n_frames):
    """Running-average value from the reduced frame sum: mean over frames."""
    return image_sum * (1.0 / float(n_frames))


class DeviceImage:
    """A torch-owned accumulation image bound to a Raytracer (rt_bind_accum), so the same memory is
    what the kernel accumulates into and what RCCL reduces."""

    def __init__(self, rt, device):
        f32 = bool(rt.flags & 0x1)
        self.tensor = torch.zeros((rt.height, rt.width, 4), dtype=torch.float32 if f32 else torch.uint8, device=device)
        rt.bind_accum(self.tensor.data_ptr(), self.tensor.numel() * self.tensor.element_size())
        self.rt = rt

    def zero_(self):
        self.tensor.zero_()
