#!/usr/bin/env python3
"""bench.py -- the headline benchmark: Mrays/s and ms/frame at 1920x1080, 8 bounces (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one frame: one pass of the path-tracing hot path (raytrace.frag::main on every pixel of
the 1920x1080 image, u_maxDepth = 8, 1 sample/pixel, fresh u_seed per frame) accumulated into the
resident float4 framebuffer.  Frames are issued --frames-in-flight B at a time (default 16 per rank) through
glrtx_render_frames: ONE launch of the render kernel (pt_render_wgwf, the workgroup-local wavefront)
covers B consecutive frames and adds their samples to the accumulator in frame order, so the result
is bit-identical to B separate launches (tests/test_gpu_parity.py) while the GPU stays full across
frame boundaries.  K steps are therefore ceil(K / B) launches; B = 1 gives one launch per frame, and
the JSON line also carries that figure ("one_launch_per_frame"), measured after the timed region.
With N > 1 ranks the image rows are sharded in interleaved 16-row stripes (one process per GPU,
global pixel coordinates, no data-path collective) and every launch ends with the RCCL all_gather of
the finished rows ("gather the framebuffer"), inside the timed region.
Scene and accumulators are resident in HBM before timing starts.  Rays are counted exactly (one
execution of intersect() in the reference's algorithm = one ray; the few the kernel can
resolve without a traversal are reported separately as rays_untraced_per_frame) by an untimed pass over the same seeds with the
counting variant of the kernel; the timed launches use the clean kernel.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- HBM roofline of the render kernel from ALGORITHMIC bytes / measured launch time
  cpu_baseline -- the CPU restatement (oracle/, "port") timed on this box's host cores (N = 1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "opengl-raytracer_amd" / "python"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
STRIPE = 16


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--config", default="headline", help="headline | c2 | c3 | c4 | c5 (parity-test configs)")
    ap.add_argument("--bvh", default="default", help="default (the config's CPU SAH tree) | lbvh (linear BVH built on the GPU, glrtx_build_lbvh)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work for the cpu_baseline sample")
    ap.add_argument("--no-gather", action="store_true", help="skip the framebuffer gather (N > 1)")
    ap.add_argument("--no-single", action="store_true", help="skip the extra one-launch-per-frame measurement (profiling runs)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames per launch of the render kernel (1 = one launch per frame; default 16 x N ranks: "
                         "16 full frames' worth of paths in flight on every GPU, whatever share of the rows it owns)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as td

    from glrt_amd import device, dist, host, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    scene, params = scenes.CONFIGS[args.config]()
    W, H = params["width"], params["height"]
    n_tri = int(scene["tri"].shape[0])

    dev = device.Device(local_rank)
    if args.bvh == "lbvh":  # BASELINE config 5: "linear-BVH traversal"
        nodes, depth, build_ms = dev.build_lbvh(scene["vert"], scene["tri"])
        scene = dict(scene, bvh=nodes, bvh_depth=depth, bvh_kind=f"lbvh, built on the GPU in {build_ms:.2f} ms")
    dev.upload_scene(scene)
    dev.set_partition(rank, world, STRIPE)
    dev.resize(W, H)
    ys = dist.owned_rows(rank, world, STRIPE, H)
    pad_rows = dist.max_owned_rows(world, STRIPE, H)
    # the accumulator lives in a torch tensor so that RCCL can gather it; the kernel writes it in place
    accum = torch.zeros((pad_rows, W, 4), dtype=torch.float32, device="cuda")
    dev.bind_accum(accum.data_ptr(), W * 16)
    stream = torch.cuda.current_stream()
    dev.set_stream(stream.cuda_stream)

    def seed(f):
        return host.frame_seed(f)

    B = args.frames_in_flight if args.frames_in_flight > 0 else min(16 * world, 256)

    def run(f0, f1, gather=True, per_launch=None):
        """Frames [f0, f1): per_launch frames per launch of the render kernel, the framebuffer gathered after every launch."""
        per_launch = per_launch or B
        img = accum
        for g in range(f0, f1, per_launch):
            n = min(per_launch, f1 - g)
            if n == 1:
                dev.render(dict(params, seed=seed(g)))
            else:
                dev.render_frames(params, [seed(g + i) for i in range(n)])
            if world > 1 and gather and not args.no_gather:
                img = dist.gather_rows(accum, H, STRIPE)
        return img

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    # ---- exact ray count for the timed frames (untimed, counting kernel variant)
    dev.count_rays(True)
    dev.reset_stats()
    run(args.warmup, args.warmup + args.steps, gather=False)
    dev.sync()
    rays_local = int(dev.stats().rays)
    untraced_local = int(dev.stats().rays_untraced)
    dev.count_rays(False)
    accum.zero_()
    dev.reset_stats()

    # ---- warm-up, then K timed frames
    run(0, args.warmup)
    dev.sync()
    dev.reset_stats()
    barrier()
    t0 = time.perf_counter()
    dev.timer_begin()
    img = run(args.warmup, args.warmup + args.steps)
    ev_ms = dev.timer_end()
    barrier()
    t1 = time.perf_counter()
    dev.sync()
    st = dev.stats()
    del img

    # ---- the same frames once more, one launch per frame (reported next to the headline figure)
    single = None
    if B > 1 and not args.no_single:
        n1 = min(args.steps, 20)
        run(0, 2, per_launch=1)
        barrier()
        t2 = time.perf_counter()
        run(args.warmup, args.warmup + n1, per_launch=1)
        barrier()
        single = torch.tensor([time.perf_counter() - t2], dtype=torch.float64, device="cuda")
        if world > 1:
            td.all_reduce(single, op=td.ReduceOp.MAX)
        single = float(single.item()) / n1

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    rays = torch.tensor([rays_local, untraced_local], dtype=torch.int64, device="cuda")
    kern_ms = torch.tensor([st.kernel_ms_total / max(st.kernel_launches, 1)], dtype=torch.float64, device="cuda")
    if world > 1:
        td.all_reduce(elapsed, op=td.ReduceOp.MAX)
        td.all_reduce(rays, op=td.ReduceOp.SUM)
        td.all_reduce(kern_ms, op=td.ReduceOp.MAX)
    elapsed_s, total_rays, total_untraced, kernel_ms = float(elapsed.item()), int(rays[0].item()), int(rays[1].item()), float(kern_ms.item())

    # ---- roofline of the render kernel (per launch, per GPU): algorithmic bytes / measured launch duration
    scene_b = scenes.scene_bytes(scene)
    frames_per_launch = st.launches / max(st.kernel_launches, 1)
    # per frame 16 B read + 16 B write per owned pixel, times the frames one launch covers, + one read of the compact scene
    algo_bytes = int(len(ys) * W * 32 * frames_per_launch) + scene_b
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    traffic = None
    pmc = ROOT / "profiles" / "r01_pmc_traffic.json"
    if pmc.exists() and args.config == "headline" and world == 1:
        try:
            prof = json.loads(pmc.read_text())  # measured per launch of prof["frames_per_launch"] frames; scaled to this run's launches
            traffic = prof["hbm_bytes_per_launch"] / prof.get("frames_per_launch", 1) * frames_per_launch
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "kernel": "pt_render_wgwf<false, false>", "kernel_ms_avg": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": algo_bytes, "frames_per_launch": round(frames_per_launch, 3),
                "note": "branchy scalar-FP32 traversal: VALU/latency-bound, not HBM-bound (DESIGN.md section 6)"}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pt_oracle
        cores = effective_cpus(pt_oracle.max_threads())
        acc = np.zeros((H, W, 4), np.float32)
        t = time.perf_counter()
        _, r = pt_oracle.render(scene, dict(params, seed=seed(args.warmup)), accum=acc, threads=cores)
        one = time.perf_counter() - t
        n = int(min(max(args.cpu_seconds / max(one, 1e-3), 1), 64))
        cpu_rays, t = 0, time.perf_counter()
        for i in range(n):
            _, r = pt_oracle.render(scene, dict(params, seed=seed(args.warmup + i)), accum=acc, threads=cores)
            cpu_rays += r
        dt = time.perf_counter() - t
        cpu_baseline = {"value": round(cpu_rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                        "sample": f"{n} full frames of the same workload (same seeds as the first timed frames), "
                                  f"{dt:.1f} s, OpenMP over rows", "ms_per_frame": round(dt / n * 1e3, 2)}

    if rank == 0:
        out = {
            "metric": "Mrays/s at 1920x1080, 8 bounces" if args.config == "headline" else f"Mrays/s ({args.config})",
            "value": round(total_rays / elapsed_s / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_s / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {n_tri} triangles (BVH {scene['bvh_kind']}), {W}x{H}, "
                                   f"u_maxDepth={params['max_depth']}, {params['n_samples']} spp/frame",
                       "partition": f"{world} x interleaved {STRIPE}-row stripes" + ("" if world == 1 else
                                    (", no gather" if args.no_gather else ", RCCL all_gather of the framebuffer after every launch")),
                       "frames_in_flight": B,
                       "one_launch_per_frame": None if single is None else
                           {"ms_per_step": round(single * 1e3, 4), "value": round(total_rays / args.steps / single / 1e6, 3)},
                       "rays_per_frame": round(total_rays / args.steps, 1),
                       # of those, shadow rays whose light test cannot change the radiance (both outcomes bit-identical):
                       # counted like the reference counts them, resolved without a traversal
                       "rays_untraced_per_frame": round(total_untraced / args.steps, 1),
                       "mpaths_per_s": round(W * H * params["n_samples"] * args.steps / elapsed_s / 1e6, 3),
                       "event_ms_per_step": round(ev_ms / args.steps, 4)},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out), flush=True)

    dev.set_stream(0)
    dev.bind_accum(0, 0)
    dev.close()
    if world > 1:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
