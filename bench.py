#!/usr/bin/env python3
"""bench.py -- the headline benchmark: Mrays/s and ms/frame at 1920x1080, 8 bounces (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            (N > 1: spawns its own N ranks, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

The workload is the reference's progressive accumulation loop (Window::mainloop calling render() once per
frame with a fresh u_seed, window.cpp:121-169, :226-252) on the 1920x1080 image at u_maxDepth = 8, 1 sample
per pixel and frame, every frame added to the resident float4 accumulator.

One STEP = one pass of the hot path over one batch: with N ranks a step is N consecutive frames of that loop,
the image rows sharded over the ranks in interleaved 8-row stripes (one process per GPU, global pixel
coordinates, no data-path collective).  Every GPU therefore traces one full frame's worth of paths
(2,073,600) per step whatever N is -- "scaling": "weak"; at N = 1 a step is exactly one frame.  The JSON line
reports ms_per_step and config.ms_per_frame = ms_per_step / N.

Frames are issued through glrtx_render_frames: ONE launch of the render kernel (pt_render_wgwf, the
workgroup-local wavefront) covers a batch of consecutive frames and adds their samples to the accumulator in
frame order, bit-identical to separate launches (tests/test_gpu_parity.py).  The K steps are cut into
ceil(K / --steps-per-launch) launches of (nearly) equal size, so a K that is not a multiple of the batch does
not leave an under-filled remainder launch.  With N > 1 the finished rows are gathered to rank 0 over RCCL
ONCE, at the end of the timed region (the accumulators stay resident on their GPUs in between, SURVEY.md
8(e)); --gather-every L gathers after every L-th launch instead.

Scene and accumulators are resident in HBM before timing starts.  Rays are counted exactly by an untimed pass
over the same seeds with the counting variant of the kernel; the timed launches use the clean kernel.  `value`
counts the rays that were actually TRAVERSED; the reference's algorithm additionally executes intersect() for
shadow rays whose light test cannot change the radiance (config.rays_reference_equivalent_per_frame).

`python bench.py --gpus N` without WORLD_SIZE in the environment re-launches itself as N ranks under
torch.distributed.run (child process; the parent never touches the GPU) and exits with the child's status.

Before timing, the accumulator the counting kernel produced for the timed steps is compared, bit for bit, with an untimed replay of the
same steps by the kernel that is timed (the compilation without ray counting): a mismatch aborts the run.

Rank 0 prints ONE JSON line (contract in the task statement) with these extra objects:
  roofline      -- HBM roofline of the render kernel from ALGORITHMIC bytes / measured launch time (what north_star asks for)
  roofline_vmem -- what the CUs' vector-memory pipes can take of this kernel's instruction mix at the kernel's own clock, and the TA busy counter
  roofline_valu -- vector-ALU issue roofline of the same kernel at the same clock, with an estimate of the SIMDs' whole issue time
  roofline_aux  -- the two kernels that ARE HBM-bound: accumulate_planes_kernel and resolve_kernel, GB/s against the HBM peak
  cpu_baseline  -- the CPU restatement (oracle/, "port") timed on this box's host cores (N = 1 only)
and in config: `strong` (N > 1: a fixed --steps-per-launch (48) frames in flight in total, the framebuffer gathered after every launch) and `predicted`
(N = 1: what rank 0's share of an N-rank step costs on this one GPU, for N = 2, 4, 8 -- the compute side of the curve, no gather).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import pathlib
import socket
import subprocess
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "opengl-raytracer_amd" / "python"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
N_CU, N_SIMD = 256, 1024
# The two issue rooflines of the render kernel are priced PER CLOCK and multiplied by the clock the render kernel itself ran at -- GRBM_GUI_ACTIVE / 8 / duration
# of the profiled dispatches (profiles/r04_pmc_summary.json: clock_ghz), NOT the nominal 2.4 GHz and not a wall-clock rate of another kernel (round 3 did both).
# The micro-benchmarks behind the per-clock figures run as >= 2 s trains and print the clock measured IN their kernels (s_memtime / s_memrealtime):
# 2.36-2.41 GHz, the clock the render kernel holds too (2.38), so their wall-clock rates are directly comparable as well.
# A SIMD with two or more resident waves issues one wave64 v_fma / v_add / v_mul per 1.96-1.98 clk (profiles/r04_ubench_valu.txt; min/max/compare/select/DPP
# forms 3.15 as a pure stream, a scalar ALU instruction 3.2, a taken branch 5.9).
VALU_CLK_PER_INST = 1.97
# What a CU's vector-memory pipe (address unit + L1 front end) takes per wave-level load, clk per instruction per CU from the SPAN of a saturated kernel
# (tools/ubench/ta.hip; profiles/r04_ubench_ta_records.txt, r04_ubench_ta_sustained.txt; TA_TA_BUSY reads 0.965-0.98 over that span):
#   every lane its own 64-byte record, 52 lanes in ~25 records -- the node fetch of a traversal step with one record per lane: 37.0
#   the two lanes of a pair share a record -- the pair-cooperative node fetch:                                               24.5
#   unit-stride / one line per quad -- the state, ray-record and plane streams:                                               16.1 (the pipe's floor)
# (Round 3's 16.8 / 11 clk were in-wave timings that assumed all 16 waves of a CU run side by side for the whole kernel; a wave's loop covers two thirds of
#  the kernel's span, so the pipe's capacity is the span figure: 25.4 clk for 64 lanes in 64 lines, 24.1 G/s chip-wide at 2.38 GHz, as the wall clock says.)
VMEM_CLK_NODE_LANE, VMEM_CLK_NODE_PAIR, VMEM_CLK_STREAM = 37.0, 24.5, 16.1
STRIPE = 8  # rows per stripe: 1080 rows over 8 ranks = 136 / 128 rows per rank (16-row stripes: 144 / 128, 6.7 % off balance)
STEPS_PER_LAUNCH = 48  # frames' worth of paths in flight on every GPU per launch (frames in flight; round 3, ms per frame: 8: 1.127, 24: 1.060, 48: 1.043; 2.4 GB of path state and sample planes at 1080p)

# Test hook: tests/ replace this with a factory of CPU renderers (same interface as GpuRenderer) to rehearse the
# N > 1 control flow under gloo.  The product path never sets it.
RENDERER_FACTORY = None


def effective_cpus(omp_max: int) -> int:
    """Host cores this process may actually use: affinity mask and cgroup CPU quota, capped by OpenMP's view."""
    n = omp_max
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = pathlib.Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def kernel_source_sha16() -> str:
    """Identifies the device code a committed profile belongs to: hash of the kernel sources and the build flags."""
    import hashlib
    h = hashlib.sha256()
    pkg = ROOT / "opengl-raytracer_amd"
    for f in sorted((pkg / "csrc").glob("*")) + [pkg / "Makefile"]:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def launch_plan(steps: int, per_launch: int):
    """Cut `steps` into ceil(steps / per_launch) launches of nearly equal size: [(first_step, n_steps), ...]."""
    if steps <= 0:
        return []
    n = -(-steps // max(per_launch, 1))
    base, extra = divmod(steps, n)
    plan, s = [], 0
    for i in range(n):
        k = base + (1 if i < extra else 0)
        plan.append((s, k))
        s += k
    return plan


def spawn_ranks(n: int) -> int:
    """Parent of a self-launched N-rank run: start torch.distributed.run as a CHILD and return its exit status.
    Nothing in this process has touched HIP (no torch.cuda call, no libglrtx call)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


class GpuRenderer:
    """One rank's share of the image on one MI355X through the C ABI (libglrtx.so); there is no CPU fallback."""

    def __init__(self, rank, world, local_rank, scene, params, bvh):
        import torch
        from glrt_amd import device, dist, host
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        self.torch = torch
        self.dev = device.Device(local_rank)
        self.scene = scene
        if bvh == "lbvh":  # BASELINE config 5: "linear-BVH traversal"
            nodes, depth, build_ms = self.dev.build_lbvh(scene["vert"], scene["tri"])
            nodes, lf = host.lights_first(nodes, scene["tri"], scene["mat"])  # as glrt::Scene::parse does after any builder
            self.scene = dict(scene, bvh=nodes, bvh_depth=depth, bvh_kind=f"lbvh, built on the GPU in {build_ms:.2f} ms", bvh_lights_first=lf)
        elif bvh == "sah-gpu":  # binned SAH by levels + exact sweep below, built on the device (glrtx_build_bvh_sah)
            self.dev.build_bvh_sah(scene["vert"], scene["tri"])  # (the first build allocates)
            nodes, depth, build_ms = self.dev.build_bvh_sah(scene["vert"], scene["tri"])
            nodes, lf = host.lights_first(nodes, scene["tri"], scene["mat"])
            self.scene = dict(scene, bvh=nodes, bvh_depth=depth, bvh_kind=f"SAH by levels, built on the GPU in {build_ms:.2f} ms", bvh_lights_first=lf)
        elif bvh == "reference":  # the reference host's own tree: its builder restated rule for rule (glrt_bvh_build_reference), left in its own child order
            from glrt_amd import scenes as _scenes
            self.scene = _scenes.rebuild_bvh(scene, "reference")
            self.scene["bvh_kind"] = "the reference host's own tree (bvh.cpp:72-160 restated: one axis binned, never re-ordered)"
        elif bvh in ("sah-reinsert", "sah-hits", "sah-reinsert-hits"):
            # the CPU SAH tree, optionally + insertion-based optimisation (glrt_bvh_reinsert), optionally + the children of every fork ordered by the closest hits of a
            # calibration frame (glrtx_hit_histogram: one 480x270 frame of this camera counted by the render kernel; glrt_bvh_order_by_hits)
            from glrt_amd import scenes as _scenes
            self.scene = _scenes.rebuild_bvh(scene, "sah-reinsert" if "reinsert" in bvh else "sah")
            self.scene["bvh_kind"] = "sah + reinsertion (CPU)" if "reinsert" in bvh else "sah"
            if bvh.endswith("-hits"):
                self.dev.upload_scene(self.scene)
                self.dev.resize(480, 270)
                hist = self.dev.hit_histogram(dict(params, width=480, height=270, seed=host.frame_seed(12345)), self.scene["tri"].shape[0])
                self.scene["bvh"], n_ex = host.order_by_hits(self.scene["bvh"], hist, self.scene["tri"], self.scene["mat"])
                self.scene["bvh_kind"] += f" + children ordered by the hits of a calibration frame ({n_ex} forks exchanged)"
        W, H = params["width"], params["height"]
        self.params = params
        self.dev.upload_scene(self.scene)
        self.dev.set_partition(rank, world, STRIPE)
        self._part = (rank, world)
        self.dev.resize(W, H)
        pad_rows = dist.max_owned_rows(world, STRIPE, H)
        # the accumulator lives in a torch tensor so that RCCL can gather it; the kernel writes it in place
        self.accum = torch.zeros((pad_rows, W, 4), dtype=torch.float32, device="cuda")
        self.dev.bind_accum(self.accum.data_ptr(), W * 16, pad_rows)
        # ONE stream for the render launches, torch's own kernels on the accumulator (zero_, clone, index_select) and the point at which RCCL picks
        # the rows up: a torch stream of our own, made current.  (torch's default stream has handle 0, which glrtx_set_stream reads as "use the
        # context's own non-blocking stream": launches there would not be ordered with torch's work or with the collective.)
        self.stream = torch.cuda.Stream(device=local_rank)
        torch.cuda.set_stream(self.stream)
        self.dev.set_stream(self.stream.cuda_stream)
        assert self.stream.cuda_stream != 0
        self.device = torch.device("cuda", local_rank)

    def render_frames(self, f0, n, seed_of):
        if n == 1:
            self.dev.render(dict(self.params, seed=seed_of(f0)))
        else:
            self.dev.render_frames(self.params, [seed_of(f0 + i) for i in range(n)])

    def prepare_frames(self, f0, n, seed_of):
        """Everything the host can do for a launch ahead of time -- the params struct, the seed array -- so that issuing it is one C call.  (A device that has been
        idle runs the next launch slower: +2 % after 1 ms, +5 % after 3 ms, +9 % after 10 ms, profiles/r04_ab_launch_warmth.txt; the timed region starts behind a
        barrier, so what the host does between the barrier and the launch is idle time of the device.)"""
        from glrt_amd import device
        if n == 1:
            return (1, device.make_params(dict(self.params, seed=seed_of(f0))), None)
        sd = np.ascontiguousarray(np.asarray([seed_of(f0 + i) for i in range(n)], np.float32).reshape(-1, 2))
        return (n, device.make_params(dict(self.params, seed=(0.0, 0.0))), sd)

    def render_prepared(self, prep):
        n, p, sd = prep
        if n == 1:
            self.dev.render(p)
        else:
            self.dev.render_frames(p, sd)

    def device_sync(self):
        self.torch.cuda.synchronize()

    def count_rays(self, on):
        self.dev.count_rays(on)

    def sync(self):
        self.dev.sync()

    def stats(self):
        return self.dev.stats()

    def reset_stats(self):
        self.dev.reset_stats()

    def timer_begin(self):
        self.dev.timer_begin()

    def timer_end(self):
        return self.dev.timer_end()

    def predict_weak(self, world, steps_per_launch, seed_of, launches=3):
        """ms per step of rank 0's share of a `world`-rank weak-scaling step, on this one GPU (no gather): stripes s % world == 0, world x
        steps_per_launch frames per launch.  Leaves the partition as it was."""
        ms = []
        try:
            self.dev.set_partition(0, world, STRIPE)
            for it in range(launches):
                self.dev.reset_stats()
                self.dev.render_frames(self.params, [seed_of(it * world * steps_per_launch + f) for f in range(world * steps_per_launch)])
                self.dev.sync()
                st = self.dev.stats()
                ms.append((st.kernel_ms_total + st.accumulate_ms_total) / steps_per_launch)
        finally:
            self.dev.set_partition(0, 1, STRIPE)
        return sorted(ms[1:])[len(ms[1:]) // 2]

    def predict_strong(self, world, frames_in_flight, seed_of, launches=3):
        """ms per FRAME of rank 0's share of a `world`-rank strong-scaling launch, on this one GPU (no gather): stripes s % world == 0 of `frames_in_flight`
        frames per launch -- 1 / world of the paths a one-GPU launch holds.  Leaves the partition as it was."""
        ms = []
        try:
            self.dev.set_partition(0, world, STRIPE)
            for it in range(launches):
                self.dev.reset_stats()
                self.dev.render_frames(self.params, [seed_of(it * frames_in_flight + f) for f in range(frames_in_flight)])
                self.dev.sync()
                st = self.dev.stats()
                ms.append((st.kernel_ms_total + st.accumulate_ms_total) / frames_in_flight)
        finally:
            self.dev.set_partition(0, 1, STRIPE)
        return sorted(ms[1:])[len(ms[1:]) // 2]

    def time_one_rank(self, f0, n, seed_of, per_launch):
        """Seconds this GPU ALONE takes for frames [f0, f0 + n) as a one-rank partition, `per_launch` frames in flight -- the N = 1 run of the strong-scaling
        workload, measured inside an N-rank run (the other ranks wait) so that the line carries its own denominator.  HIP events on the render stream around the
        launches, behind an untimed launch of the same shape (buffers sized, device warm).  Leaves partition and accumulator binding as they were."""
        torch = self.torch
        W, H = self.params["width"], self.params["height"]
        rank, world = self._part
        full = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        self.dev.sync()
        self.dev.bind_accum(0, 0, 0)
        try:
            self.dev.set_partition(0, 1, STRIPE)
            self.dev.bind_accum(full.data_ptr(), W * 16, H)
            plan = [self.prepare_frames(f0 + a, k, seed_of) for a, k in launch_plan(n, per_launch)]
            warm = self.prepare_frames(f0, min(n, per_launch), seed_of)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.render_prepared(warm)
            e0.record(self.stream)
            for prep in plan:
                self.render_prepared(prep)
            e1.record(self.stream)
            self.dev.sync()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e-3
        finally:
            self.dev.bind_accum(0, 0, 0)
            self.dev.set_partition(rank, world, STRIPE)
            self.dev.bind_accum(self.accum.data_ptr(), W * 16, self.accum.shape[0])

    def render_full_reference(self, f0, n, seed_of, per_launch=16):
        """Frames [f0, f0 + n) rendered by THIS GPU alone as a one-rank partition into a fresh full-size accumulator (untimed; what the gathered image of
        an N-rank run of the same frames must equal bit for bit).  Leaves partition and accumulator binding as they were."""
        torch = self.torch
        W, H = self.params["width"], self.params["height"]
        rank, world = self._part
        full = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        self.dev.sync()
        self.dev.bind_accum(0, 0, 0)
        try:
            self.dev.set_partition(0, 1, STRIPE)
            self.dev.bind_accum(full.data_ptr(), W * 16, H)
            for a, k in launch_plan(n, per_launch):
                self.render_frames(f0 + a, k, seed_of)
            self.dev.sync()
        finally:
            self.dev.bind_accum(0, 0, 0)
            self.dev.set_partition(rank, world, STRIPE)
            self.dev.bind_accum(self.accum.data_ptr(), W * 16, self.accum.shape[0])
        return full

    def use_own_stream(self, on):
        """Render launches on the context's own stream (where consecutive calls feed one running launch) or back on the torch stream everything else here
        is ordered on.  The device is idle at both switches."""
        self.dev.sync(); self.torch.cuda.synchronize()
        self.dev.set_stream(0 if on else self.stream.cuda_stream)

    def resolve_ms(self):
        """Device time of one resolve pass over the owned rows (screen.frag), ms: (per launch in a train of 32 launches between one pair of events, a single launch
        between two events).  The second also measures the command processor's latency on both sides of a 12-us kernel."""
        self.dev.resolve_rgba8(2.2, True)
        self.dev.resolve_rgba8(2.2, True)
        single = float(self.dev.stats().resolve_ms_last)
        return float(self.dev.resolve_burst_ms(2.2, 32)), single

    def close(self):
        self.dev.set_stream(0)
        self.dev.bind_accum(0, 0, 0)
        self.dev.close()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--config", default="headline", help="headline | c2 | c3 | c4 | c5 (parity-test configs)")
    ap.add_argument("--bvh", default="default", help="default (the config's CPU SAH tree) | lbvh (linear BVH built on the GPU, glrtx_build_lbvh) | sah-gpu (binned SAH built on the GPU, glrtx_build_bvh_sah) | sah-reinsert (the CPU SAH tree + glrt_bvh_reinsert) | sah-hits, sah-reinsert-hits (+ every fork's children ordered by the hits of a calibration frame: glrtx_hit_histogram, glrt_bvh_order_by_hits) | reference (the reference host's own tree, its builder restated: glrt_bvh_build_reference)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work for the cpu_baseline sample")
    ap.add_argument("--no-llvmpipe", action="store_true", help="skip the llvmpipe leg of cpu_baseline (the repository's own GLSL port through oracle/glref)")
    ap.add_argument("--llvmpipe-frames", type=int, default=3, help="timed llvmpipe frames (a 1080p headline frame takes seconds there)")
    ap.add_argument("--no-gather", action="store_true", help="skip the framebuffer gather (N > 1)")
    ap.add_argument("--gather-every", type=int, default=0, help="gather the framebuffer to rank 0 after every L-th launch (0: once, at the end of the timed region)")
    ap.add_argument("--no-single", action="store_true", help="skip the extra one-launch-per-frame measurement (profiling runs)")
    ap.add_argument("--steps-per-launch", "--frames-in-flight", dest="steps_per_launch", type=int, default=STEPS_PER_LAUNCH,
                    help="steps per launch of the render kernel (at N ranks a step is N frames, so a launch covers N x this many frames: "
                         "this many full frames' worth of paths in flight on every GPU; 1 = one step per launch)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals: tests/, or a 1-GPU box)")
    ap.add_argument("--device-map", default="", help="rehearsal aid: comma-separated HIP ordinal per local rank (e.g. 0,0 runs two ranks on one GPU; "
                                                     "RCCL needs distinct GPUs, so combine with --backend gloo --no-gather)")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: before torch.cuda / libglrtx are touched in this process
        raise SystemExit(spawn_ranks(args.gpus))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as td

    from glrt_amd import dist, host, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                         f"or without WORLD_SIZE set (bench.py then spawns its ranks itself)")

    scene, params = scenes.CONFIGS[args.config]()
    W, H = params["width"], params["height"]
    n_tri = int(scene["tri"].shape[0])

    factory = RENDERER_FACTORY or GpuRenderer
    dev_ord = int(args.device_map.split(",")[local_rank]) if args.device_map else local_rank
    R = factory(rank, world, dev_ord, scene, params, args.bvh)
    scene = R.scene
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": R.device} if args.backend == "nccl" else {}
        td.init_process_group(args.backend, rank=rank, world_size=world, **kw)
    ranks_seen = td.get_world_size() if world > 1 else 1       # what the process group itself reports after init ("nccl" IS RCCL on ROCm)
    backend_seen = td.get_backend() if world > 1 else None
    ys = dist.owned_rows(rank, world, STRIPE, H)
    gatherer = dist.RowGather(world, STRIPE, H, R.accum) if world > 1 else None  # index tensors built once, here

    def seed(f):
        return host.frame_seed(f)

    S = max(1, args.steps_per_launch)

    def prepare(s0, n_steps, per_launch=None):
        """The launches of steps [s0, s0 + n_steps), with everything the host can do ahead of time done (renderers without prepare_frames: the plain plan)."""
        plan = launch_plan(n_steps, per_launch or S)
        if not hasattr(R, "prepare_frames"):
            return [(a, k, None) for a, k in plan]
        return [(a, k, R.prepare_frames((s0 + a) * world, k * world, seed)) for a, k in plan]

    gather_sink = None  # a list while the timed region runs: (event, event) or seconds per collective of THIS rank

    def gather_now(local):
        """The framebuffer gather, timed when the timed region asks for it: HIP events on the render stream around the collective and its de-interleave (what
        lies between them is the collective alone: the first event fires when the render work in front of it has finished)."""
        if gather_sink is None:
            return gatherer.gather_to_root(local)
        if local.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            img = gatherer.gather_to_root(local)
            e1.record()
            gather_sink.append((e0, e1))
        else:
            t = time.perf_counter()
            img = gatherer.gather_to_root(local)
            gather_sink.append(time.perf_counter() - t)
        return img

    def run(s0, n_steps, per_launch=None, gather=True, prepared=None):
        """Steps [s0, s0 + n_steps) = frames [s0*N, (s0+n_steps)*N), in balanced launches; returns the gathered image (rank 0) or None."""
        img = None
        plan = prepared if prepared is not None else [(a, k, None) for a, k in launch_plan(n_steps, per_launch or S)]
        for i, (a, k, prep) in enumerate(plan):
            if prep is not None:
                R.render_prepared(prep)
            else:
                R.render_frames((s0 + a) * world, k * world, seed)
            last = i + 1 == len(plan)
            if gatherer is not None and gather and not args.no_gather and (last or (args.gather_every > 0 and (i + 1) % args.gather_every == 0)):
                img = gather_now(R.accum)
        return img

    def clear_accum():
        """Zero the resident accumulator, ordered on the render stream, without making the host wait (a renderer that keeps a copy elsewhere clears that too)."""
        R.accum.zero_()
        if hasattr(R, "cleared"):
            R.cleared()

    def barrier():
        R.device_sync()
        if world > 1:
            td.barrier()
        R.device_sync()

    # ---- untimed preamble, enqueued WITHOUT a host wait from its first launch to the barrier in front of the timed region:
    #   (1) the timed steps by the counting kernel variant: the exact ray count;
    #   (2) the same steps by the kernel that is timed (compiled without ray counting): its accumulator must equal (1)'s bit for bit -- compared on the device, the
    #       verdict read after the timed region (a mismatch aborts the run before anything is printed);
    #   (3) the W warm-up steps.
    # Why without a wait: a launch issued on a device that has been idle runs longer, and the effect outlasts the idle time by tens of milliseconds -- 20 frames take
    # 20.2-20.7 ms back to back, +2 % behind a pause of 1 ms, +5 % behind 3 ms, +9 % behind 10 ms, +12 % from cold (tools/gpu_launch_warmth.py,
    # profiles/r04_ab_launch_warmth.txt).  With a host round trip between (1) and (2), as in rounds 1-3, the driver's cadence (K = 20, W = 5) timed its one launch
    # 5-6 % above the steady state of the same launch; with W = 20 it did not.  All launches are prepared beforehand so that issuing one is a single C call.
    # torch's own kernels used below (copy, compare, reduce, fill) are launched once HERE, on the idle device: the first launch of a kernel loads its code object, and
    # such a load stalls the device for milliseconds in the middle of whatever is enqueued (profiles/r04_ab_pipeline_robustness.txt) -- inside the preamble that is an
    # idle gap in front of the timed region (measured: the timed launch of 20 frames 21.4 ms the first time the process runs these ops, 20.6 ms the second time)
    _w = R.accum.clone()
    _ = (R.accum.view(torch.int32) == _w.view(torch.int32)).all().to(torch.int32).reshape(1)
    R.accum.zero_()
    del _w, _
    if gatherer is not None and not args.no_gather:
        gatherer.gather_to_root(R.accum)  # the collective's and the de-interleave's kernels as well (N > 1), and the communicator is set up
    barrier()
    count_launches = prepare(args.warmup, args.steps)
    replay_launches = prepare(args.warmup, args.steps)
    warm_launches = prepare(0, args.warmup)
    timed_launches = prepare(args.warmup, args.steps)
    R.reset_stats()
    clear_accum()
    R.count_rays(True)
    run(args.warmup, args.steps, gather=False, prepared=count_launches)
    R.count_rays(False)
    counted_img = R.accum.clone()
    clear_accum()
    run(args.warmup, args.steps, gather=False, prepared=replay_launches)
    same = (R.accum.view(torch.int32) == counted_img.view(torch.int32)).all().to(torch.int32).reshape(1)
    del counted_img
    clear_accum()
    # no collector pauses inside anything timed below: a full collection of a process that has imported torch takes ~50 ms on this host -- nothing for a launch that is
    # already enqueued, 1.5 ms per frame for a leg of 32 calls with a sync behind each (seen in config 5's line: profiles/r06_launch_shapes.txt)
    gc.collect()
    gc.disable()
    run(0, args.warmup, prepared=warm_launches)
    R.sync()  # the first host wait since the preamble began; it is part of the barrier that opens the timed region
    st0 = R.stats()
    rays_local = int(st0.rays)
    untraced_local = int(st0.rays_untraced)
    barrier()
    gather_sink = []
    t0 = time.perf_counter()
    R.timer_begin()
    img = run(args.warmup, args.steps, prepared=timed_launches)
    t_issued = time.perf_counter()
    ev_ms = R.timer_end()
    barrier()
    t1 = time.perf_counter()
    gather_ms_local = sum((g[0].elapsed_time(g[1]) if isinstance(g, tuple) else g * 1e3) for g in gather_sink)
    n_gathers = len(gather_sink)
    gather_sink = None
    R.sync()
    st1 = R.stats()
    # what the timed region added to the context's running totals
    import types
    st = types.SimpleNamespace(kernel_ms_total=st1.kernel_ms_total - st0.kernel_ms_total, kernel_launches=st1.kernel_launches - st0.kernel_launches,
                               launches=st1.launches - st0.launches, accumulate_ms_total=getattr(st1, "accumulate_ms_total", 0.0) - getattr(st0, "accumulate_ms_total", 0.0),
                               node_fetch_last=getattr(st1, "node_fetch_last", 0))
    del img
    if world > 1:
        td.all_reduce(same, op=td.ReduceOp.MIN)
    if int(same.item()) != 1:
        raise SystemExit("bench.py: the timed kernel's image differs from the counting kernel's image for the same steps: refusing to report a time for it")

    # ---- the same steps once more, one frame per launch (reported next to the headline figure; N = 1 only)
    # Round 6: on the context's OWN stream -- where nothing but library calls can order work against the accumulator -- such a burst of calls runs as one fed launch (the
    # calls behind the first publish their frames to the launch that is already running: glrtx.hip feed_append); with GLRTX_NO_FEED=1 every call is a launch of its own,
    # overlapped with its neighbours (round 3), which is also what a render-resolve-save loop and a caller's stream get.  Both are measured, the same frames each.
    single, single_overlapped, single_feed, single_synced, single_overlapped_slots, single_synced_stats = None, None, None, None, {}, {}
    if world == 1 and S > 1 and not args.no_single:
        n1 = max(args.steps, 48)  # (a burst: its one ramp and one drain are spread over this many frames)
        own = hasattr(R, "use_own_stream")
        if own:
            R.use_own_stream(True)
        try:
            for leg in ("fed", "overlapped"):
                if leg == "overlapped":
                    if not own or "GLRTX_NO_FEED" in os.environ:
                        continue
                    os.environ["GLRTX_NO_FEED"] = "1"
                try:
                    # untimed: every internal slot has its buffers.  Fed launches alternate between three slots and take their frames' planes 16 at a
                    # time on demand (0.5 GB per hipMalloc at 1080p, ~1 ms each): three bursts of the timed length leave nothing to allocate inside the
                    # timed one (an 8-frame warm-up did, until round 6's last session: 2.2 ms of allocations in a 48-frame burst)
                    for _ in range(3):
                        run(0, n1, per_launch=1)
                        barrier()
                    st_a = R.stats()
                    t2 = time.perf_counter()
                    run(args.warmup, n1, per_launch=1)
                    barrier()
                    dt = (time.perf_counter() - t2) / n1
                    st_b = R.stats()
                finally:
                    if leg == "overlapped":
                        del os.environ["GLRTX_NO_FEED"]
                if leg == "fed":
                    single = dt
                    single_feed = {"kernel_launches": int(st_b.kernel_launches - st_a.kernel_launches),
                                   "frames_appended_to_a_running_launch": int(getattr(st_b, "feed_appended", 0) - getattr(st_a, "feed_appended", 0))}
                else:
                    single_overlapped = dt
                    single_overlapped_slots = {"pipe_slots": int(getattr(st_b, "pipe_slots", 0)), "pipe_resident_max": int(getattr(st_b, "pipe_resident_max", 0)),
                                               "kernel_launches": int(st_b.kernel_launches - st_a.kernel_launches)}
            # ... and with a sync behind every call: a host that looks at every frame before it asks for the next one.  Every launch is then alone on the device (and is
            # shaped for its tail: glrtx.hip, launch_wgwf `shape`)
            n_sync = min(n1, 32)
            barrier()
            R.sync()
            st_a = R.stats()
            t2 = time.perf_counter()
            t_call = 0.0
            for i in range(n_sync):
                t3 = time.perf_counter()
                run(args.warmup + i, 1, per_launch=1)
                t_call += time.perf_counter() - t3
                barrier()
            single_synced = (time.perf_counter() - t2) / n_sync
            R.sync()
            st_b = R.stats()
            single_synced_stats = {"kernel_ms_per_launch": round((st_b.kernel_ms_total - st_a.kernel_ms_total) / n_sync, 4), "wf_state_mib": int(getattr(st_b, "wf_state_mib", 0)),
                                   "accumulate_ms_per_launch": round((getattr(st_b, "accumulate_ms_total", 0.0) - getattr(st_a, "accumulate_ms_total", 0.0)) / n_sync, 4),
                                   "host_ms_in_the_call": round(t_call * 1e3 / n_sync, 4),
                                   "feed_launches": int(getattr(st_b, "feed_launches", 0) - getattr(st_a, "feed_launches", 0))}
        finally:
            if own:
                R.use_own_stream(False)

    # ---- strong scaling (N > 1): the SAME K frames as a one-GPU run of K steps -- S frames in flight in total per launch, i.e. S / N
    # frames' worth of paths on every GPU -- and the framebuffer gathered to rank 0 after EVERY launch
    strong_s, strong_err = None, None
    if world > 1:
        # The gather of launch i runs while launch i + 1 renders: the rows are snapshotted (a device-to-device copy on the render stream), the
        # collective is issued asynchronously on the snapshot (RCCL's stream waits for the copy), and the next launch goes on writing the
        # resident accumulator meanwhile; the snapshot is reused only after its collective has completed.
        def run_strong(f0, n_frames):
            img, work, snap = None, None, None
            for a, k in launch_plan(n_frames, S):
                R.render_frames(f0 + a, k, seed)
                if gatherer is not None and not args.no_gather:
                    if work is not None:
                        img = gatherer.finish(work)
                    if snap is None:
                        snap = torch.empty_like(R.accum)
                    snap.copy_(R.accum)
                    work = gatherer.gather_to_root_async(snap)
            if work is not None:
                img = gatherer.finish(work)
            return img
        try:
            run_strong(0, min(S, args.steps))
            barrier()
            t2 = time.perf_counter()
            img = run_strong(args.warmup * world, args.steps)
            barrier()
            strong_s = time.perf_counter() - t2
            del img
            R.sync()
        except Exception as e:  # the weak figure is the contract line: never lose it to the extra measurement
            strong_err = f"{type(e).__name__}: {e}"
            strong_s = None

    # ---- the strong-scaling DENOMINATOR, measured in this same run (N > 1): rank 0's GPU alone renders the same K frames as a one-rank partition with S frames in
    # flight per launch -- exactly what `bench.py --gpus 1 --steps K` times -- while the other ranks wait at the barrier behind it.
    n1_s, n1_err = None, None
    if world > 1 and hasattr(R, "time_one_rank"):
        if rank == 0:
            try:
                n1_s = R.time_one_rank(args.warmup * world, args.steps, seed, S)
            except Exception as e:  # an extra: never fatal -- and never a reason for rank 0 to miss the barrier the other ranks are waiting in
                n1_err = f"{type(e).__name__}: {e}"
        barrier()

    # ---- self-check of what the collective delivered (N > 1, untimed): min(K, 4) of the timed steps are rendered again by all ranks into cleared
    # accumulators and gathered over the same collective path -- once as the weak region issues them, once as the strong region does -- and rank 0
    # compares each gathered image, bit for bit, with the same frames rendered by its own GPU alone as a one-rank partition.  (The timed regions drop
    # their images; without this the first multi-GPU run would time a gather nobody looked at.)
    gather_check, strong_check = None, None
    if world > 1 and gatherer is not None and not args.no_gather:
        def pixels_differing(img, ref):
            return int((img.contiguous().view(torch.int32) != ref.contiguous().view(torch.int32)).any(dim=-1).sum().item())
        try:
            nchk = min(args.steps, 4)
            f_lo, f_n = args.warmup * world, nchk * world
            R.sync()
            R.accum.zero_()
            R.reset_stats()
            img_w = run(args.warmup, nchk)
            barrier()
            img_w = img_w.clone() if img_w is not None else None
            R.accum.zero_()
            R.reset_stats()
            img_s = run_strong(f_lo, f_n) if strong_err is None else None
            barrier()
            if rank == 0:
                try:  # (rank 0 alone: whatever happens here, it must still reach the barrier the other ranks are waiting in)
                    ref = R.render_full_reference(f_lo, f_n, seed)
                    d = pixels_differing(img_w, ref)
                    gather_check = "bit-identical" if d == 0 else f"{d} of {W * H} pixels differ"
                    if img_s is not None:
                        d = pixels_differing(img_s, ref)
                        strong_check = "bit-identical" if d == 0 else f"{d} of {W * H} pixels differ"
                    del ref
                except Exception as e:
                    gather_check = f"error: {type(e).__name__}: {e}"
            barrier()
        except Exception as e:  # reported, never fatal to the contract line
            gather_check = f"error: {type(e).__name__}: {e}"

    # ---- what rank 0's share of an N-rank weak step costs on this GPU, N = 2, 4, 8 (N = 1 runs on a GPU only)
    predicted, resolve_ms = None, None
    try:
        if world == 1 and hasattr(R, "predict_weak") and not args.no_single:
            predicted = {}
            for w in (2, 4, 8):
                ms = R.predict_weak(w, S, seed)
                predicted[str(w)] = {"ms_per_step": round(ms, 4)}
                if hasattr(R, "predict_strong"):
                    predicted[str(w)]["strong_ms_per_frame"] = round(R.predict_strong(w, S, seed), 4)
        resolve_ms = R.resolve_ms() if (world == 1 and hasattr(R, "resolve_ms")) else None
    except Exception as e:  # extras only
        predicted = {"error": f"{type(e).__name__}: {e}"}

    dev0 = R.accum.device
    elapsed = torch.tensor([t1 - t0, strong_s if strong_s is not None else -1.0], dtype=torch.float64, device=dev0)
    rays = torch.tensor([rays_local, untraced_local], dtype=torch.int64, device=dev0)
    kern_ms = torch.tensor([st.kernel_ms_total / max(st.kernel_launches, 1)], dtype=torch.float64, device=dev0)
    # per rank: [render-kernel ms summed over the timed launches, event-timed ms of its collectives in the timed region]
    per_rank = torch.zeros((world, 2), dtype=torch.float64, device=dev0)
    per_rank[rank, 0] = st.kernel_ms_total
    per_rank[rank, 1] = gather_ms_local
    n1 = torch.tensor([n1_s if n1_s is not None else -1.0], dtype=torch.float64, device=dev0)
    if world > 1:
        td.all_reduce(elapsed, op=td.ReduceOp.MAX)
        td.all_reduce(rays, op=td.ReduceOp.SUM)
        td.all_reduce(kern_ms, op=td.ReduceOp.MAX)
        td.all_reduce(per_rank, op=td.ReduceOp.SUM)
        td.all_reduce(n1, op=td.ReduceOp.MAX)
    per_rank = per_rank.cpu().tolist()
    n1_s = float(n1.item()) if float(n1.item()) > 0 else None
    elapsed_s, ref_rays, total_untraced, kernel_ms = float(elapsed[0].item()), int(rays[0].item()), int(rays[1].item()), float(kern_ms.item())
    strong_s = float(elapsed[1].item()) if float(elapsed[1].item()) > 0 and strong_err is None else None  # (MAX over ranks: -1 everywhere = not measured)
    traced_rays = ref_rays - total_untraced
    n_frames = args.steps * world

    gc.enable()

    # ---- rooflines of the render kernel (per launch, per GPU)
    scene_b = scenes.scene_bytes(scene)
    frames_per_launch = st.launches / max(st.kernel_launches, 1)
    # HBM: per frame 16 B read + 16 B write per owned pixel, times the frames one launch covers, + one read of the compact scene
    algo_bytes = int(len(ys) * W * 32 * frames_per_launch) + scene_b
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    # PMC counters cannot be collected inside this run (rocprofv3 wraps the process): the per-frame instruction counts and HBM-side bytes
    # come from the committed profile of this same command (tools/profile.sh + tools/summarize_profile.py -> profiles/rNN[_config]_pmc_summary.json) and are scaled to the
    # frames' worth of pixels a launch covers on this GPU.  The summary records a hash of the kernel sources it was taken from; if the
    # sources have changed since, the derived figures are reported as null ("stale_profile") instead of looking measured.
    traffic, valu, vmem, prof, stale = None, None, None, None, None
    # the newest committed summary of this config whose kernel-source hash matches this build (profiles/rNN[_cfg]_pmc_summary.json)
    import re
    cands = []
    for f in (ROOT / "profiles").glob("r*_pmc_summary.json"):
        m = re.fullmatch(r"r(\d+)([a-z]?)(?:_(c\d|headline))?_pmc_summary\.json", f.name)
        if m and (m.group(3) or "headline") == args.config:
            cands.append(((int(m.group(1)), m.group(2)), f))
    for _, f in sorted(cands, reverse=True):
        try:
            pj = json.loads(f.read_text())
        except Exception:
            continue
        pj["file"] = f"profiles/{f.name}"
        if pj.get("kernel_source_sha16") == kernel_source_sha16():
            prof, stale = pj, None
            break
        if stale is None:
            stale = f"{pj['file']} was taken from kernel sources {pj.get('kernel_source_sha16')}, this build is {kernel_source_sha16()}"
    # which tree the committed counters of this config were taken under (tools/profile.sh's third argument; profiles/README.md), against this run's
    profiled_bvh = {"c5": "sah-gpu"}.get(args.config, "default")
    tree_note = None if args.bvh == profiled_bvh else (
        f"the committed counters were taken under --bvh {profiled_bvh}, this run's tree is --bvh {args.bvh}: instructions and bytes per frame follow the steps per ray "
        "(the CPU and the device SAH trees: within 1 % of each other; reinserted trees: ~4 % fewer; the reference host's own tree: a third MORE on the Cornell scenes, "
        "profiles/r06_reference_tree.txt), so the issue rooflines and `traffic` of this line are that approximate")
    if prof is not None and prof.get("clock_ghz"):
        frames_equiv = frames_per_launch * len(ys) / H
        sec = kernel_ms * 1e-3
        clock = float(prof["clock_ghz"])
        traffic = prof["hbm_bytes_per_frame"] * frames_equiv
        vi = prof["valu_insts_per_frame"] * frames_equiv
        rate = vi / sec / 1e9 if sec > 0 else 0.0
        valu_peak = N_SIMD * clock / VALU_CLK_PER_INST
        si = prof.get("salu_insts_per_frame", 0.0) * frames_equiv
        bi = prof.get("branch_insts_per_frame", 0.0) * frames_equiv
        # issue time of everything a SIMD issues besides vector memory: vector 1.97 clk (a floor: a third of the step's vector instructions are 3.15-clk forms),
        # scalar 3.2, branches 3.2 (not taken) .. 5.9 (taken)
        issue_lo = (vi * VALU_CLK_PER_INST + si * 3.2 + bi * 3.2) / (N_SIMD * clock * 1e9 * sec) if sec > 0 else 0.0
        issue_hi = (vi * 2.4 + si * 3.2 + bi * 5.9) / (N_SIMD * clock * 1e9 * sec) if sec > 0 else 0.0
        valu = {"bound": "valu-issue", "achieved": round(rate, 2), "peak": round(valu_peak, 1), "unit": "G wave-instructions/s",
                "frac": round(rate / valu_peak, 4), "clock_ghz": round(clock, 4), "lane_util": prof.get("lane_util"),
                "effective_fp32_lane_frac": None if prof.get("lane_util") is None else round(rate / valu_peak * prof["lane_util"], 4),
                "simd_issue_busy_estimate": [round(issue_lo, 3), round(issue_hi, 3)],
                "wave_cycles_waiting_frac": prof.get("wave_cycles_waiting_frac"),
                "valu_insts_per_launch": int(vi), "salu_insts_per_launch": int(si), "branch_insts_per_launch": int(bi), "source": prof["file"], "profiled_tree": tree_note or "this run's",
                "note": "SQ_INSTS_VALU per frame from the committed rocprofv3 --pmc pass of this command / kernel time measured in this run; peak = 1024 SIMDs x the clock "
                        "the profiled kernel ran at (GRBM_GUI_ACTIVE / 8 / duration) / 1.97 clk per wave64 v_fma/v_add/v_mul (profiles/r04_ubench_valu.txt, measured at "
                        "the same in-kernel clock); simd_issue_busy_estimate adds the scalar and branch instructions the SIMDs issue beside them; lane_util = "
                        "SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU / 64, calibrated 1.000 / 0.500 on full / half exec masks"}
        if prof.get("vmem_insts_per_frame"):
            mi = prof["vmem_insts_per_frame"] * frames_equiv
            mrate = mi / sec / 1e9 if sec > 0 else 0.0
            nf = int(getattr(st, "node_fetch_last", 0))  # 0 one record per lane, 1 pair-cooperative, 2 the two in alternate steps
            node = min(prof.get("node_fetch_insts_per_frame", 0.0), prof["vmem_insts_per_frame"]) * frames_equiv
            share = node / mi if mi > 0 else 0.0
            clk_node = (VMEM_CLK_NODE_LANE, VMEM_CLK_NODE_PAIR, 0.5 * (VMEM_CLK_NODE_LANE + VMEM_CLK_NODE_PAIR))[nf]
            clk_mix = share * clk_node + (1.0 - share) * VMEM_CLK_STREAM   # clk of the pipe per instruction of this kernel's mix
            vmem_peak = N_CU * clock / clk_mix
            vmem = {"bound": "vmem-issue", "achieved": round(mrate, 3), "peak": round(vmem_peak, 2), "unit": "G wave-level vector-memory instructions/s",
                    "frac": round(mrate / vmem_peak, 4), "clock_ghz": round(clock, 4), "node_fetch": ("one record per lane", "pair-cooperative", "pair-cooperative and one record per lane in alternate steps")[nf],
                    "node_fetch_share_of_insts": round(share, 4), "clk_per_inst_of_the_mix": round(clk_mix, 2),
                    "clk_per_inst": {"node fetch": clk_node, "streams": VMEM_CLK_STREAM},
                    "ta_busy_counter": None if prof.get("ta_busy_frac") is None else round(prof["ta_busy_frac"], 4),
                    # the same ratio for the PROFILED dispatches (their own duration): what ta_busy_counter is to be compared with -- `frac` moves with this run's
                    # launch shape (a launch of few frames carries more of the launch's fixed cost per frame)
                    "frac_of_profiled_dispatches": None if not prof.get("kernel_ms_per_frame_profiled") else
                        round(prof["vmem_insts_per_frame"] / (prof["kernel_ms_per_frame_profiled"] * 1e-3) / 1e9 / vmem_peak, 4),
                    # the hardware's own figure carried over to this run: the profiled dispatches' busy counter times (their time per frame / this run's).  Since round 6
                    # the priced model (`frac`) reads 0.09-0.11 above the counter (0.01-0.09 until round 5; its per-instruction prices are round 4's span measurements,
                    # taken on round 4's kernel) -- of the two, this is the one to quote
                    "frac_from_ta_busy": None if prof.get("ta_busy_frac") is None or not prof.get("kernel_ms_per_frame_profiled") else
                        round(prof["ta_busy_frac"] * prof["kernel_ms_per_frame_profiled"] / (kernel_ms / max(frames_equiv, 1e-9)), 4),
                    "vmem_insts_per_launch": int(mi), "source": prof["file"], "profiled_tree": tree_note or "this run's",
                    "note": "SQ_INSTS_VMEM_RD + SQ_INSTS_VMEM_WR per frame (committed --pmc pass) / kernel time of this run, against what a CU's vector-memory pipe takes for "
                            "this mix: node fetches (4 per wave-step, tools/gpu_travstats.py) at the span-measured cost of their access pattern, everything else at the pipe's "
                            "floor, times the kernel's own clock.  ta_busy_counter = TA_TA_BUSY_sum / 256 / (GRBM_GUI_ACTIVE / 8) of the profiled dispatches (0.965-0.98 on the "
                            "saturated micro-benchmark).  Reading (DESIGN.md section 6): with one record per lane the pipe is the busiest unit of the chip, ~0.8; the "
                            "SIMDs' issue is ~0.7-0.8 beside it; waves spend half their cycles in s_waitcnt.  Neither is saturated: a wave's step is a serial chain of "
                            "fetch, wait and ~160 instructions at >= 4.5 clk each, and four waves per SIMD overlap those chains to ~80 % of the busier unit"}
    roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_measured_in_run": False,
                "traffic_source": None if traffic is None else f"{prof['file']} (separate --pmc FETCH_SIZE / WRITE_SIZE passes of this command, scaled by this run's launch; {prof.get('traffic_note', '')})",
                "stale_profile": stale,
                "kernel": "pt_render_wgwf<false, %s, %d>" % ("true" if scene.get("bvh_kind", "").startswith("chain") else "false", int(getattr(st, "node_fetch_last", 0))),
                "kernel_ms_avg": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": algo_bytes, "frames_per_launch": round(frames_per_launch, 3),
                "note": "a per-lane BVH gather with branchy scalar FP32: not HBM-bound (DESIGN.md section 6); the units that are busy are the CUs' vector-memory "
                        "pipes and the SIMDs' instruction issue: roofline_vmem, roofline_valu"}
    # the two kernels of the path that ARE HBM-bound, timed in this run by HIP events on the launch stream
    aux = None
    if world == 1 and RENDERER_FACTORY is None:
        aux = {}
        n_planes = frames_per_launch * params["n_samples"]
        if st.kernel_launches and st.accumulate_ms_total > 0 and n_planes > 1:
            ms = st.accumulate_ms_total / st.kernel_launches
            by = len(ys) * W * 16 * (n_planes + 2)  # every plane read once, the accumulator read and written
            aux["accumulate_planes_kernel"] = {"bound": "hbm", "achieved": round(by / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                               "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ms": round(ms, 4), "bytes": int(by),
                                               "what": f"{n_planes:g} sample planes of {W}x{len(ys)} float4 added to the accumulator"}
        if resolve_ms:
            resolve_ms, resolve_single = resolve_ms if isinstance(resolve_ms, tuple) else (resolve_ms, None)
            by = len(ys) * W * 20  # 16 B accumulator texel in, 4 B RGBA8 out (screen.frag:15-25)
            aux["resolve_kernel"] = {"bound": "hbm", "achieved": round(by / (resolve_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(by / (resolve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ms": round(resolve_ms, 4), "bytes": int(by),
                                     "single_launch_between_two_events_ms": None if resolve_single is None else round(resolve_single, 4),
                                     "what": f"{W}x{len(ys)}: rgb / count, clamp, pow(1 / 2.2), RGBA8; ms = per launch in a train of 32 launches (glrtx_debug_resolve_burst); "
                                             "a one-shot kernel over 41 MB: its ramp and drain count as much as the streaming rate (DESIGN.md section 9 row 7: a version with a third of the instructions takes the same time); "
                                             "profiles/r06_aux_kernels.json has 4K"}

    cpu_baseline, oracle_check = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and RENDERER_FACTORY is None:
        from oracle import pt_oracle
        cores = effective_cpus(pt_oracle.max_threads())
        scratch = np.zeros((H, W, 4), np.float32)
        t = time.perf_counter()
        pt_oracle.render(scene, dict(params, seed=seed(args.warmup)), accum=scratch, threads=cores)  # sizes the sample (and warms the cores); its image is not kept
        one = time.perf_counter() - t
        del scratch
        n = int(min(max(args.cpu_seconds / max(one, 1e-3), 1), 64))
        acc = np.zeros((H, W, 4), np.float32)  # the accumulation loop over the FIRST n TIMED frames, from cleared accumulators
        cpu_rays, t = 0, time.perf_counter()
        for i in range(n):
            _, r = pt_oracle.render(scene, dict(params, seed=seed(args.warmup + i)), accum=acc, threads=cores)
            cpu_rays += r
        dt = time.perf_counter() - t
        cpu_baseline = {"value": round(cpu_rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                        "sample": f"{n} full frames of the same workload (same seeds as the first timed frames), "
                                  f"{dt:.1f} s, OpenMP over rows; counts every intersect() execution of the reference's algorithm",
                        "ms_per_frame": round(dt / n * 1e3, 2)}
        # the oracle as the CHECKER of what was timed: the same n frames by the GPU (untimed, the timed kernel, frames in flight) into a cleared accumulator,
        # compared bit for bit with the image the oracle just accumulated.  A difference aborts the run like the counting-kernel check does.
        gpu_img = R.render_full_reference(args.warmup, n, seed, per_launch=S).cpu().numpy()
        diff = int((gpu_img.view(np.uint32) != acc.view(np.uint32)).any(-1).sum())
        if diff:
            raise SystemExit(f"bench.py: the GPU's accumulator differs from the oracle's on {diff} of {W * H} pixels over timed frames "
                             f"[{args.warmup}, {args.warmup + n}): refusing to report a time for it")
        oracle_check = f"bit-identical over {n} frames"
        del gpu_img, acc
        # the llvmpipe leg (BASELINE.json: "next to the reference timed through Mesa llvmpipe on the box's own host cores in the same run"): the reference's shader
        # file does not travel, so this is the repository's own GLSL statement of it (oracle/glsl/pt_port.frag; bit-identical to the reference's images on llvmpipe in
        # the build container, tests/test_glsl_port.py), run through oracle/glref in a CHILD process (llvmpipe's LLVM stays out of the process that holds the GPU).
        lp = None
        if one > 40.0:  # (config 3's brute-force scan: a frame takes the CPU port a minute and llvmpipe about twice that)
            lp = f"skipped: a frame of this config takes the CPU port {one:.0f} s, llvmpipe about twice that"
        elif not args.no_llvmpipe:
            try:
                cmd = [sys.executable, "-m", "oracle.glport", "--config", args.config, "--frames", str(args.llvmpipe_frames if one < 5.0 else 1), "--warmup", "1",
                       "--first-frame", str(args.warmup), "--check"]
                r = subprocess.run(cmd, cwd=str(ROOT), capture_output=True, text=True, timeout=900)
                lp = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else {"available": False, "reason": f"child exited {r.returncode}: {r.stderr[-300:]}"}
            except Exception as e:
                lp = {"available": False, "reason": f"{type(e).__name__}: {e}"}
            if lp.get("available"):
                lp["value"] = round(lp["rays"] / lp["frames"] / (lp["ms_per_frame"] * 1e-3) / 1e6, 3)
                lp["unit"] = "Mrays/s"
            else:
                lp = "absent on box: " + str(lp.get("reason"))
        cpu_baseline["llvmpipe"] = lp

    if world == 1:
        scaling_strong = {"value": round(traced_rays / elapsed_s / 1e6, 3), "unit": "Mrays/s", "ms_per_frame": round(elapsed_s / n_frames * 1e3, 4),
                          "speedup_vs_n1_predicted": 1.0,
                          "predicted_from_one_gpu": None if not predicted or "error" in predicted else
                              {n: {"ms_per_frame": v.get("strong_ms_per_frame"),
                                   "speedup_vs_n1_predicted": None if not v.get("strong_ms_per_frame") else round(elapsed_s / n_frames * 1e3 / v["strong_ms_per_frame"], 3)}
                               for n, v in predicted.items()},
                          "note": "N = 1: strong and weak coincide.  predicted_from_one_gpu = rank 0's share (1/N of the rows, S frames in flight) of an N-rank strong launch "
                                  "timed on this GPU, kernel + plane accumulation, no gather: the compute side of the strong curve"}
    else:
        strong_ms = None if strong_s is None else strong_s / args.steps * 1e3
        n1_ms = None if n1_s is None else n1_s / args.steps * 1e3
        scaling_strong = {"value": None if strong_s is None else round(traced_rays / world / strong_s / 1e6, 3), "unit": "Mrays/s",
                          "ms_per_frame": None if strong_ms is None else round(strong_ms, 4),
                          "n1_same_run": None if n1_ms is None else {"ms_per_frame": round(n1_ms, 4), "value": round(traced_rays / world / n1_s / 1e6, 3),
                                                                     "what": f"the same {args.steps} frames on rank 0's GPU alone, {S} frames in flight, HIP events"},
                          "speedup_vs_n1_predicted": None if (strong_ms is None or n1_ms is None) else round(n1_ms / strong_ms, 3),
                          "error": strong_err or n1_err,
                          "note": "the same K frames whatever N is; speedup = this run's own one-GPU time of those frames / the N-GPU time (the driver's N = 1 line is the "
                                  "authoritative denominator; this one makes the line readable by itself)"}
    # ---- which figure is the line's `value` (round 6).  BASELINE.json's metric is Mrays/s of ONE 1920x1080 frame size at 1, 2, 4, 8 GPUs (">= 6x at 8 GPUs"): the
    # same K frames whatever N is -- strong scaling.  Until round 5 `value` at N > 1 was the weak figure (N frames per step: it grows with N by construction) and the
    # strong one sat beside it; now the strong region IS the line: value = rays of K frames / the barrier-bracketed, max-over-ranks time of those K frames, S frames in
    # flight in total per launch, the framebuffer gathered to rank 0 after every launch.  The weak figure moves to config.weak.  At N = 1 the two coincide.  If the
    # strong region failed (config.strong.error) the line falls back to the weak figure and says so in `scaling`.
    weak_value = traced_rays / elapsed_s / 1e6
    weak_ms_per_step = elapsed_s / args.steps * 1e3
    strong_is_line = world == 1 or strong_s is not None
    line_value = weak_value if (world == 1 or strong_s is None) else traced_rays / world / strong_s / 1e6
    line_ms_per_step = weak_ms_per_step if (world == 1 or strong_s is None) else strong_s / args.steps * 1e3
    env_overrides = {k: v for k, v in sorted(os.environ.items()) if k.startswith(("GLRTX_", "GLRT_"))}
    st_final = R.stats()
    shadow_search = {0: "exact", 1: "range-limited"}.get(int(getattr(st_final, "shadow_limited", -1)), "unknown")
    if rank == 0:
        gather_note = "" if world == 1 else (", no gather" if args.no_gather else
                                             (f", RCCL gather of the framebuffer to rank 0 every {args.gather_every} launches and at the end of the timed region"
                                              if args.gather_every > 0 else ", RCCL gather of the framebuffer to rank 0 once, at the end of the timed region"))
        out = {
            "metric": "Mrays/s at 1920x1080, 8 bounces" if args.config == "headline" else f"Mrays/s ({args.config})",
            "value": round(line_value, 3),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(line_ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong" if strong_is_line else "weak",
            # `value` above is the STRONG figure (see line_value): the SAME K frames whatever N is.  scaling_strong repeats it with its own one-GPU denominator, measured
            # on rank 0's GPU alone IN THIS RUN; config.weak holds the weak figure (N frames per step) that was `value` until round 5.
            "scaling_strong": scaling_strong,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {n_tri} triangles (BVH {scene['bvh_kind']}; the light side first at {scene.get('bvh_lights_first', 0)} forks), {W}x{H}, "
                                   f"u_maxDepth={params['max_depth']}, {params['n_samples']} spp/frame",
                       "step": ("1 frame of the accumulation loop (the same K frames whatever N is), its rows sharded over the N GPUs in interleaved stripes"
                                if (world > 1 and strong_is_line) else
                                f"{world} consecutive frame(s) of the accumulation loop = one full frame's worth of pixels per GPU"),
                       "frames_per_step": 1 if strong_is_line else world,
                       "ms_per_frame": round(line_ms_per_step if strong_is_line else elapsed_s / n_frames * 1e3, 4),
                       # which shadow-ray search the device ran (glrtx_stats.shadow_limited: the exact search is the default, the range limit an opt-in outside the
                       # bit-exact contract) and every GLRTX_* / GLRT_* variable set in this process: a line taken with a switch thrown says so
                       "shadow_search": shadow_search,
                       "env_overrides": env_overrides,
                       # the weak-scaling figure (N frames per step: every GPU renders one full frame's worth of pixels per step whatever N is; one gather at the end
                       # of the timed region) -- `value` until round 5
                       "weak": None if world == 1 else {"value": round(weak_value, 3), "unit": "Mrays/s", "ms_per_step": round(weak_ms_per_step, 4), "frames_per_step": world,
                                                        "ms_per_frame": round(elapsed_s / n_frames * 1e3, 4)},
                       "partition": f"{world} x interleaved {STRIPE}-row stripes" + gather_note,
                       "steps_per_launch": S,
                       "launches": [k for _, k in launch_plan(args.steps, S)],
                       "one_launch_per_frame": None if single is None else
                           {"ms_per_step": round(single * 1e3, 4), "value": round(traced_rays / args.steps / single / 1e6, 3), "frames": max(args.steps, 48),
                            "what": "glrtx_render once per frame, back to back on the context's own stream, one sync at the end (window.cpp:121-169's cadence without its per-frame "
                                    "read-back): the calls behind the first feed the launch that is already running",
                            **(single_feed or {}),
                            "overlapped_launches": None if single_overlapped is None else
                                {"ms_per_step": round(single_overlapped * 1e3, 4), "value": round(traced_rays / args.steps / single_overlapped / 1e6, 3), **single_overlapped_slots,
                                 "what": "the same calls with GLRTX_NO_FEED=1: one launch per call, overlapped with its neighbours (round 3) -- what a loop that resolves "
                                         "every frame, or a caller's stream, gets"},
                            "synced_launches": None if single_synced is None else
                                {"ms_per_step": round(single_synced * 1e3, 4), "value": round(traced_rays / args.steps / single_synced / 1e6, 3), "frames": min(max(args.steps, 48), 32), **single_synced_stats,
                                 "what": "a glrtx_sync behind every call: every launch alone on the device, wall time per frame"}},
                       "rays_per_frame": round(traced_rays / n_frames, 1),
                       # the reference's algorithm also executes intersect() for shadow rays whose light test cannot change the
                       # radiance (both outcomes bit-identical); those are resolved without a traversal and NOT part of `value`
                       "rays_reference_equivalent_per_frame": round(ref_rays / n_frames, 1),
                       "rays_untraced_per_frame": round(total_untraced / n_frames, 1),
                       "mrays_per_s_reference_equivalent": round(ref_rays / n_frames / (line_ms_per_step / (1 if strong_is_line else world) * 1e-3) / 1e6, 3),
                       "mpaths_per_s": round(W * H * params["n_samples"] / (line_ms_per_step / (1 if strong_is_line else world) * 1e-3) / 1e6, 3),
                       "event_ms_per_step": round(ev_ms / args.steps, 4),
                       "host_ms_to_issue_timed_launches": round((t_issued - t0) * 1e3, 3),
                       # N > 1: the image RCCL gathered for min(K, 4) of the timed steps against the same frames rendered by rank 0's GPU alone
                       "gather_check": gather_check if world > 1 else None,
                       # what torch.distributed reported after init_process_group (backend "nccl" IS RCCL on ROCm), and per rank: the render kernel's time over the
                       # timed launches and the event-timed collectives (+ de-interleave on the root) of the timed region
                       "rccl_ranks_seen": ranks_seen, "backend_seen": backend_seen,
                       "per_rank": [{"rank": i, "kernel_ms": round(k, 3), "gather_ms": round(g, 3)} for i, (k, g) in enumerate(per_rank)],
                       "gathers_in_timed_region": n_gathers,
                       "gather_check_what": None if world == 1 else f"frames [{args.warmup * world}, {(args.warmup + min(args.steps, 4)) * world}) re-rendered untimed by all ranks, gathered to "
                                                                      "rank 0 over the timed region's collective path, compared bit for bit with a one-rank render of the same frames on rank 0's GPU",
                       "timed_kernel_image_check": f"bit-identical to the counting kernel's accumulator over the {args.steps} timed steps (untimed replay)",
                       # N = 1: the GPU's accumulator over the first n timed frames against the oracle's (the CPU restatement, pinned by the llvmpipe fixtures)
                       "oracle_image_check": oracle_check,
                       "run_to_run": "contexts of one binary differ by +-0.5 % since the path state is sized by the launch (round 6, profiles/README.md: r06_lone_knobs; 3 % until round 5 with where a 4.6-GB "
                                     "path-state buffer landed physically, profiles/r04_context_regimes.txt); box to box the driver's cadence spans 0.96-1.015 ms per frame (profiles/r06_run_to_run.txt)",
                       # strong scaling: the same K frames whatever N is, S frames in flight IN TOTAL per launch, framebuffer gathered after every launch
                       "strong": ({"error": strong_err} if strong_err else None) if strong_s is None else
                           {"frames": args.steps, "frames_in_flight_total": S, "gather": "none" if args.no_gather else "to rank 0 after every launch, overlapped with the next launch",
                            "ms_per_frame": round(strong_s / args.steps * 1e3, 4), "gather_check": strong_check,
                            "value": round(traced_rays / world / strong_s / 1e6, 3), "unit": "Mrays/s",
                            "note": "rays of K frames / elapsed; compare with the N = 1 line's value for strong-scaling efficiency"},
                       # N = 1: rank 0's share of an N-rank weak step on this GPU (kernel + plane accumulation, no gather, RCCL not involved)
                       "predicted": None if predicted is None else
                           (predicted if "error" in predicted else
                            {n: dict(v, value=round(traced_rays / args.steps * int(n) / (v["ms_per_step"] * 1e-3) / 1e6, 1)) for n, v in predicted.items()})},
            "roofline": roofline,
            "roofline_vmem": vmem,
            "roofline_valu": valu,
            "roofline_aux": aux,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out), flush=True)

    R.close()
    if world > 1:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
