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
