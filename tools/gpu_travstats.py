"""Traversal statistics of a config (libglrtx_stats.so, -DGLRTX_TRAV_STATS: the C++ statement of the step with counters): wave-steps, lanes and cache lines per step, steps per
ray; GLRTX_TRAVSTATS_JSON=path writes them as JSON (tools/profile.sh -> profiles/<tag>_travstats.json, which bench.py's roofline_vmem prices the node fetches with).
    python tools/gpu_travstats.py [config = headline] [frames]
"""
import sys, ctypes as C; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
import os, json
device.lib_path = lambda: device.LIB_DIR / os.environ.get("GLRTX_STATS_LIB", "libglrtx_stats.so")
cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
sc, pr = scenes.CONFIGS[cfg]()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"]); d.count_rays(True)
L = device.lib()
for v in (2,):
    d.set_variant(v); d.reset_stats()
    out = (C.c_ulonglong * 8)(); L.glrtx_debug_trav_stats(out)
    tr = (C.c_ulonglong * 4)(); L.glrtx_debug_trav_trips(tr)
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    d.render_frames(pr, [host.frame_seed(f) for f in range(B)]); d.sync()
    L.glrtx_debug_trav_stats(out); o = list(out); rays = d.stats().rays
    L.glrtx_debug_trav_trips(tr); t = list(tr)
    print(f"stepping trips {t[0]}: {t[1] / max(t[0], 1):.1f} lanes with a ray at their start; {100.0 * t[2] / max(t[0], 1):.1f} % of the trips after the workgroup's queue ran out (drain), {t[3] / max(t[2], 1):.1f} lanes at their start")
    hh = (C.c_ulonglong * 16)(); L.glrtx_debug_trav_hist(hh); print("iterations histogram (<=1,2,4,8,...):", list(hh))
    sph = (C.c_ulonglong * 16)(); L.glrtx_debug_trav_sp_hist(sph); sph = list(sph); tot = max(sum(sph), 1)
    print("lane-steps by stack pointer at the step's start (0..14, 15+):", sph)
    print("  share of lane-steps with sp >= k: " + ", ".join(f"{k}: {100.0 * sum(sph[k:]) / tot:.3f} %" for k in range(4, 16)))
    print(f"rays traced {sum(hh)} of {rays} reference rays ({100.0*sum(hh)/max(rays,1):.1f} %)")
    print(f"variant {v}: rays {rays} wave_iters {o[0]} lane_iters {o[1]} simd_eff {o[1]/(64*o[0]):.3f} fork_lane {o[2]} leaf_lane {o[3]} mixed_iters {o[4]/o[0]:.3f} iters/ray {o[1]/rays:.1f} forks/ray {o[2]/rays:.1f} leaves/ray {o[3]/rays:.1f} distinct records/wave-iter {o[5]/o[0]:.1f} distinct 128B lines/wave-iter {o[6]/o[0]:.1f} (active lanes/wave-iter {o[1]/o[0]:.1f}) path-ray share of lane iters {o[7]/o[1]:.3f}")
    if os.environ.get("GLRTX_TRAVSTATS_JSON"):
        json.dump({"config": cfg, "frames": B, "rays": int(rays), "wave_iters": o[0], "lane_iters": o[1], "fork_lane": o[2], "leaf_lane": o[3], "mixed_iters": o[4], "records": o[5],
                   "lines": o[6], "path_ray_lane_iters": o[7], "trips": t, "iters_hist_log2": list(hh), "sp_hist": sph}, open(os.environ["GLRTX_TRAVSTATS_JSON"], "w"))
