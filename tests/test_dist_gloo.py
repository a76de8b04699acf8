"""The N>1 path on CPU: world_size-2 (and 3, ragged) gloo runs of the row-stripe partition and the
framebuffer gather (glrt_amd.dist).  The per-rank renderer here is the oracle standing in for the
device kernel -- test infrastructure only; what is under test is the partition arithmetic (global
coordinates, full windowSize) and the gather/de-interleave, which bench.py runs over RCCL."""
import json
import os
import pathlib
import socket
import subprocess
import sys

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


def test_bench_self_spawns_its_ranks_and_gathers_to_root(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: bench.main re-launches itself as 2 ranks under
    torch.distributed.run (tests/bench_rehearsal.py swaps the device renderer for the oracle on CPU tensors and the
    backend for gloo).  Checks the JSON line, that the ranks rendered the frames the step plan assigns, and that the
    image gathered to rank 0 at the end of the timed region equals a single-process render of the same frames."""
    root = pathlib.Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GLRT_REHEARSAL_OUT=str(tmp_path), OMP_NUM_THREADS="1")
    steps, warm, spl = 5, 2, 2
    r = subprocess.run([sys.executable, str(root / "tests" / "bench_rehearsal.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warm),
                        "--steps-per-launch", str(spl), "--config", "rehearsal", "--backend", "gloo", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    # round 6: at N > 1 the line's `value` IS the strong figure (the same K frames whatever N is); the weak figure (N frames per step) sits in config.weak
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["scaling"] == "strong"
    assert out["config"]["frames_per_step"] == 1 and out["config"]["launches"] == [2, 2, 1]
    assert out["value"] == out["scaling_strong"]["value"] and out["ms_per_step"] == out["scaling_strong"]["ms_per_frame"] == out["config"]["ms_per_frame"]
    weak = out["config"]["weak"]
    assert weak["frames_per_step"] == 2 and weak["value"] > 0 and abs(weak["ms_per_frame"] * 2 - weak["ms_per_step"]) < 1e-3 * weak["ms_per_step"] + 1e-4
    assert out["config"]["shadow_search"] in ("exact", "range-limited", "unknown") and isinstance(out["config"]["env_overrides"], dict)
    assert out["config"]["env_overrides"].get("GLRT_REHEARSAL_OUT") == str(tmp_path)  # every GLRTX_* / GLRT_* variable of the process is in the line
    assert out["value"] > 0 and out["cpu_baseline"] is None
    # weak and strong figures are both in the line; the one-GPU predictions only exist at N = 1 on a GPU
    strong = out["config"]["strong"]
    assert strong["frames"] == steps and strong["frames_in_flight_total"] == spl and strong["value"] > 0 and "every launch" in strong["gather"]
    assert out["config"]["predicted"] is None and "bit-identical" in out["config"]["timed_kernel_image_check"]
    assert out["roofline"]["traffic_measured_in_run"] is False and out["roofline_aux"] is None
    # round 5: the line explains itself: the ranks the process group reports, per-rank kernel / collective times, the strong figure at the top level
    # with its own one-GPU denominator, measured on rank 0 alone while rank 1 waits (the rehearsal renderer times its oracle-backed full render)
    cfg = out["config"]
    assert cfg["rccl_ranks_seen"] == 2 and cfg["backend_seen"] == "gloo" and cfg["gathers_in_timed_region"] == 1
    assert [r["rank"] for r in cfg["per_rank"]] == [0, 1] and all(r["gather_ms"] > 0 for r in cfg["per_rank"])
    ss = out["scaling_strong"]
    assert ss["value"] == strong["value"] and ss["ms_per_frame"] == strong["ms_per_frame"] and ss["error"] is None
    assert ss["n1_same_run"]["ms_per_frame"] > 0 and abs(ss["speedup_vs_n1_predicted"] - ss["n1_same_run"]["ms_per_frame"] / ss["ms_per_frame"]) < 2e-3
    # the counting pass, its replay by the timed kernel (image check), the warm-up, the timed region, and the strong-scaling region
    # (one warm launch, then `steps` FRAMES in launches of `spl` frames in total) cover these frames, in this order, on every rank
    timed = list(range(warm * 2, (warm + steps) * 2))
    strong_frames = list(range(0, min(spl, steps))) + list(range(warm * 2, warm * 2 + steps))
    # ... and the untimed self-check of the collective: min(steps, 4) of the timed steps once more through the weak region's gather and once
    # through the strong region's, each from cleared accumulators (rank 0's one-rank reference render is not a rank's share and is not logged)
    check = list(range(warm * 2, (warm + min(steps, 4)) * 2))
    want = timed + timed + list(range(0, warm * 2)) + timed + strong_frames + check + check
    assert out["config"]["gather_check"] == "bit-identical", out["config"]["gather_check"]
    assert strong["gather_check"] == "bit-identical"
    for k in range(2):
        assert np.load(tmp_path / f"frames_rank{k}.npy").tolist() == want
    # image: everything after the last clear -- the self-check's frames through the strong region's gather -- accumulated from zero, gathered to rank 0
    from glrt_amd import host
    from tests.bench_rehearsal import small_config
    sc, pr = small_config()
    ref = np.zeros((pr["height"], pr["width"], 4), np.float32)
    for f in check:
        pt_oracle.render(sc, dict(pr, seed=host.frame_seed(f)), accum=ref, threads=2)
    got = np.load(tmp_path / "gathered.npy")
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_bench_at_eight_ranks_under_gloo(tmp_path):
    """The driver's largest run shape -- `bench.py --gpus 8` -- rehearsed on CPU: eight ranks under gloo with the oracle-backed renderer, 72 rows in 8-row stripes
    (rank 0 owns two stripes, the others one: ragged blocks, padded for the collective).  The line must carry both self-checks of the collective as bit-identical,
    and the image gathered to rank 0 must equal a single-process render of the frames behind the last clear."""
    root = pathlib.Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GLRT_REHEARSAL_OUT=str(tmp_path), OMP_NUM_THREADS="1")
    steps, warm, spl, world = 3, 1, 2, 8
    r = subprocess.run([sys.executable, str(root / "tests" / "bench_rehearsal.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", str(warm),
                        "--steps-per-launch", str(spl), "--config", "rehearsal", "--backend", "gloo", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == steps and out["scaling"] == "strong" and out["value"] == out["scaling_strong"]["value"] > 0
    assert out["config"]["frames_per_step"] == 1 and out["config"]["weak"]["frames_per_step"] == world and out["config"]["launches"] == [2, 1]
    assert out["config"]["gather_check"] == "bit-identical", out["config"]["gather_check"]
    assert out["config"]["strong"]["gather_check"] == "bit-identical" and out["config"]["strong"]["frames"] == steps
    from glrt_amd import host
    from tests.bench_rehearsal import small_config
    sc, pr = small_config()
    ref = np.zeros((pr["height"], pr["width"], 4), np.float32)
    for f in range(warm * world, (warm + min(steps, 4)) * world):  # the self-check's frames through the strong region's gather, accumulated from zero
        pt_oracle.render(sc, dict(pr, seed=host.frame_seed(f)), accum=ref, threads=2)
    got = np.load(tmp_path / "gathered.npy")
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_bench_gather_check_reports_a_corrupted_gather(tmp_path):
    """The same rehearsal with a gather that delivers one wrong value on the weak region's path: the contract line is still printed and
    config.gather_check names the damage; the strong region's (asynchronous) path is untouched and stays bit-identical."""
    root = pathlib.Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", GLRT_REHEARSAL_CORRUPT_GATHER="1")
    r = subprocess.run([sys.executable, str(root / "tests" / "bench_rehearsal.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--steps-per-launch", "2", "--config", "rehearsal", "--backend", "gloo", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert out["config"]["gather_check"] == f"1 of {48 * 72} pixels differ", out["config"]["gather_check"]
    assert out["config"]["strong"]["gather_check"] == "bit-identical"


def test_bench_single_rank_flow_rehearsed_on_cpu():
    """The N = 1 control flow of bench.py -- prepared launches, the untimed preamble enqueued without a host wait (counting pass, replay by the timed kernel, image check read
    after the timed region, warm-up), stats taken as differences around the timed region -- with the oracle-backed renderer: one JSON line, exact frame order, ray count of the
    timed steps only."""
    root = pathlib.Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out_dir = pathlib.Path(os.environ.get("PYTEST_TMP", "/tmp")) / f"glrt_rehearsal_n1_{os.getpid()}"
    out_dir.mkdir(parents=True, exist_ok=True)
    env.update(OMP_NUM_THREADS="1", GLRT_REHEARSAL_OUT=str(out_dir))
    steps, warm = 3, 1
    r = subprocess.run([sys.executable, str(root / "tests" / "bench_rehearsal.py"), "--gpus", "1", "--steps", str(steps), "--warmup", str(warm),
                        "--steps-per-launch", "2", "--config", "rehearsal", "--backend", "gloo", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == steps and out["config"]["launches"] == [2, 1]
    assert out["config"]["strong"] is None and out["config"]["gather_check"] is None and "bit-identical" in out["config"]["timed_kernel_image_check"]
    timed = list(range(warm, warm + steps))
    frames = np.load(out_dir / "frames_rank0.npy").tolist()
    assert frames[:3 * steps + warm] == timed + timed + list(range(warm)) + timed, frames   # counting pass, replay, warm-up, timed region
    # `value` counts the rays of the timed steps (counted by the first pass over them)
    from glrt_amd import host
    from tests.bench_rehearsal import small_config
    sc, pr = small_config()
    rays = sum(pt_oracle.render(sc, dict(pr, seed=host.frame_seed(f)), threads=2)[1] for f in timed)
    assert abs(out["config"]["rays_reference_equivalent_per_frame"] * steps - rays) < 0.5


def test_launch_plan_is_balanced():
    import bench
    assert bench.launch_plan(20, 16) == [(0, 10), (10, 10)]
    assert bench.launch_plan(48, 16) == [(0, 16), (16, 16), (32, 16)]
    assert bench.launch_plan(5, 2) == [(0, 2), (2, 2), (4, 1)]
    assert bench.launch_plan(3, 16) == [(0, 3)] and bench.launch_plan(0, 16) == []


def test_row_gather_index_matches_owned_rows():
    for world, stripe, h in ((1, 16, 40), (2, 16, 64), (3, 16, 56), (8, 16, 1080)):
        src = dist.source_rows(world, stripe, h)
        pad = dist.max_owned_rows(world, stripe, h)
        assert sorted(src.tolist()) == sorted(r * pad + i for r in range(world) for i in range(len(dist.owned_rows(r, world, stripe, h))))
        for r in range(world):
            ys = dist.owned_rows(r, world, stripe, h)
            assert np.array_equal(src[ys], r * pad + np.arange(len(ys)))


@pytest.mark.parametrize("mode", ["1", "2"])
def test_bench_rank0_only_failures_do_not_leave_the_other_ranks_waiting(tmp_path, mode):
    """Round 5 added work that rank 0 does ALONE while the other ranks wait at a barrier (the one-GPU denominator of scaling_strong; before it, the one-rank reference render
    of the gather check).  If that work raises, rank 0 must still reach the barrier -- a missed one shifts every later collective and the first real multi-GPU run would hang.
    Mode 1: the denominator raises; mode 2: the reference render raises as well.  The run finishes, the contract line is printed, the failures are named in it."""
    root = pathlib.Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GLRT_REHEARSAL_OUT=str(tmp_path), OMP_NUM_THREADS="1", GLRT_REHEARSAL_FAIL_RANK0=mode)
    r = subprocess.run([sys.executable, str(root / "tests" / "bench_rehearsal.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--steps-per-launch", "2",
                        "--config", "rehearsal", "--backend", "gloo", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert out["n_gpus"] == 2 and out["value"] > 0
    ss = out["scaling_strong"]
    assert ss["n1_same_run"] is None and ss["speedup_vs_n1_predicted"] is None and "on purpose" in ss["error"]
    if mode == "2":
        assert out["config"]["gather_check"].startswith("error:") and "on purpose" in out["config"]["gather_check"]
    else:
        assert out["config"]["gather_check"] == "bit-identical"
