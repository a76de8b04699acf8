"""Row-stripe sharding across one-process-per-GPU ranks and the framebuffer gather.

A pixel depends only on (gl_FragCoord, u_windowSize, u_seed), the read-only scene and its own
accumulator (raytrace.frag:567-613), so ranks render disjoint interleaved stripes with global pixel
coordinates and no data-path collective.  The only exchange is the gather of finished rows
(torch.distributed gather / all_gather: RCCL over xGMI on GPUs, gloo in the CPU tests), and it is needed
only when an image is wanted: accumulators stay resident on their ranks in between.
"""
from __future__ import annotations

import numpy as np


def owned_rows(rank: int, world: int, stripe: int, height: int) -> np.ndarray:
    """Global y of each local accumulator row of `rank`, in local order (== glrtx_local_row_to_y)."""
    ys = []
    n_stripes = (height + stripe - 1) // stripe
    for s in range(rank, n_stripes, world):
        ys.extend(range(s * stripe, min((s + 1) * stripe, height)))
    return np.asarray(ys, np.int64)


def max_owned_rows(world: int, stripe: int, height: int) -> int:
    return max(len(owned_rows(r, world, stripe, height)) for r in range(world))


def source_rows(world: int, stripe: int, height: int) -> np.ndarray:
    """For every global row y: its row in the rank-major stack of padded per-rank blocks, rank * pad + local row."""
    pad = max_owned_rows(world, stripe, height)
    src = np.empty(height, np.int64)
    for r in range(world):
        ys = owned_rows(r, world, stripe, height)
        src[ys] = r * pad + np.arange(len(ys))
    return src


class RowGather:
    """Gather of the per-rank row blocks into the full image.  Everything that does not depend on the pixel values
    -- the de-interleave index (on the device) and the receive buffer -- is built ONCE, here; a gather is then one
    collective plus one index_select kernel, with no host-to-device copies on the timed path."""

    def __init__(self, world: int, stripe: int, height: int, like, group=None):
        import torch
        self.torch = torch
        self.world, self.stripe, self.height, self.group = world, stripe, height, group
        self.pad = max_owned_rows(world, stripe, height)
        assert like.shape[0] == self.pad, (like.shape, self.pad)
        self.src = torch.as_tensor(source_rows(world, stripe, height), device=like.device)
        self.stack = torch.empty((world * self.pad,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
        self.blocks = list(self.stack.view((world,) + tuple(like.shape)).unbind(0))  # contiguous views: receive in place

    def gather_to_root(self, local, dst: int = 0):
        """Rows of every rank -> the full (height, W, C) image on rank `dst`; None on the other ranks."""
        import torch.distributed as dist
        rank = dist.get_rank(self.group)
        dist.gather(local, self.blocks if rank == dst else None, dst=dst, group=self.group)
        return self.stack.index_select(0, self.src) if rank == dst else None

    def gather_to_root_async(self, local, dst: int = 0):
        """Start the gather of `local` (which must stay untouched until finish()) and return a handle for finish()."""
        import torch.distributed as dist
        rank = dist.get_rank(self.group)
        return (dist.gather(local, self.blocks if rank == dst else None, dst=dst, group=self.group, async_op=True), rank == dst)

    def finish(self, handle):
        """Complete a gather_to_root_async: the full image on the root, None elsewhere."""
        work, is_root = handle
        work.wait()
        return self.stack.index_select(0, self.src) if is_root else None

    def all_gather(self, local):
        """The full image on every rank."""
        import torch.distributed as dist
        dist.all_gather_into_tensor(self.stack, local.contiguous(), group=self.group)
        return self.stack.index_select(0, self.src)


def gather_rows(local, height: int, stripe: int, group=None):
    """all_gather the per-rank row blocks and de-interleave them into the full (height, W, C) image on every rank.

    local: torch tensor (rows_padded >= owned rows, W, C) on this rank's device; every rank must pass the
    same padded row count (max_owned_rows) -- stripes are ragged when height % (stripe*world) != 0.
    (One-shot convenience form; loops should keep a RowGather.)"""
    import torch.distributed as dist
    return RowGather(dist.get_world_size(group), stripe, height, local, group).all_gather(local)
