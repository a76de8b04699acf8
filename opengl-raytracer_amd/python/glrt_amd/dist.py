"""Row-stripe sharding across one-process-per-GPU ranks and the framebuffer gather.

A pixel depends only on (gl_FragCoord, u_windowSize, u_seed), the read-only scene and its own
accumulator (raytrace.frag:567-613), so ranks render disjoint interleaved stripes with global pixel
coordinates and no data-path collective.  The only exchange is the gather of finished rows
(torch.distributed all_gather: RCCL over xGMI on GPUs, gloo in the CPU tests).
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


def gather_rows(local, height: int, stripe: int, group=None):
    """all_gather the per-rank row blocks and de-interleave them into the full (height, W, C) image.

    local: torch tensor (rows_padded >= owned rows, W, C) on this rank's device; every rank must pass the
    same padded row count (max_owned_rows) -- stripes are ragged when height % (stripe*world) != 0.
    Returns the full image on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    pad = max_owned_rows(world, stripe, height)
    assert local.shape[0] == pad, (local.shape, pad)
    # concatenated-along-dim-0 output form: accepted by both the NCCL(RCCL) and the gloo backends
    out = torch.empty((world * pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    out = out.view((world,) + tuple(local.shape))
    full = torch.empty((height,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        ys = owned_rows(r, world, stripe, height)
        if len(ys):
            full[torch.as_tensor(ys, device=local.device)] = out[r, :len(ys)]
    return full
