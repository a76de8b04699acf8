"""The N>1 path on CPU: world_size-2 (and 3, ragged) gloo runs of the row-stripe partition and the
framebuffer gather (glrt_amd.dist).  The per-rank renderer here is the oracle standing in for the
device kernel -- test infrastructure only; what is under test is the partition arithmetic (global
coordinates, full windowSize) and the gather/de-interleave, which bench.py runs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

from glrt_amd import dist, scenes
from oracle import pt_oracle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, stripe, width, height, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc, pr = scenes.config_c1(width, height, max_depth=3, n_samples=2, subdiv=1)
        ys = dist.owned_rows(rank, world, stripe, height)
        full_local = np.zeros((height, width, 4), np.float32)
        for s0 in range(0, len(ys), stripe):  # each owned stripe is a contiguous global row range
            seg = ys[s0:s0 + stripe]
            pt_oracle.render(sc, pr, accum=full_local, rows=(int(seg[0]), int(seg[-1]) + 1), threads=2)
        pad = dist.max_owned_rows(world, stripe, height)
        local = torch.zeros((pad, width, 4), dtype=torch.float32)
        local[:len(ys)] = torch.from_numpy(full_local[ys])
        full = dist.gather_rows(local, height, stripe)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), full.numpy())
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("world,stripe,width,height", [(2, 16, 48, 64), (3, 16, 40, 56)])
def test_partitioned_render_plus_gather_equals_single_process(tmp_path, world, stripe, width, height):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, stripe, width, height, str(tmp_path)), nprocs=world, join=True)
    sc, pr = scenes.config_c1(width, height, max_depth=3, n_samples=2, subdiv=1)
    ref, _ = pt_oracle.render(sc, pr, threads=2)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert got.shape == ref.shape
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"rank {r} image differs"
