"""Tree quality on one config: traversal statistics (lib built with -DGLRTX_TRAV_STATS) and frame time per BVH builder.
    python tools/gpu_bvh_compare.py [config] [builder ...]      builders: sah lbvh lbvh-opt (CPU statements via libglrt_host)"""
import sys, os, ctypes as C; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
kinds = sys.argv[2:] or ["sah", "lbvh"]
sc0, pr = scenes.CONFIGS[cfg]()
for kind in kinds:
    sc = scenes.rebuild_bvh(sc0, kind)
    # frame time with the product library
    device._lib = None; device.lib_path = lambda: device.LIB_DIR / "libglrtx.so"
    d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
    seeds = lambda f0: [host.frame_seed(f0 + i) for i in range(8)]
    ms = []
    for r in range(5):
        d.render_frames(pr, seeds(8 * r)); d.sync(); ms.append(d.stats().kernel_ms_last / 8)
    d.close()
    # statistics with the instrumented library
    device._lib = None; device.lib_path = lambda: device.LIB_DIR / "libglrtx_stats.so"
    d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"]); d.count_rays(True)
    L = device.lib(); out = (C.c_ulonglong * 8)(); L.glrtx_debug_trav_stats(out)
    d.render_frames(pr, seeds(0)); d.sync(); L.glrtx_debug_trav_stats(out); o = list(out); rays = d.stats().rays
    d.close()
    print(f"{cfg} {kind:9s} depth {sc['bvh_depth']:3d}  {sorted(ms)[2]:.3f} ms/frame  lane-iters/ray {o[1]/rays:.2f} (forks {o[2]/rays:.2f}, leaves {o[3]/rays:.2f})  distinct lines/wave-iter {o[6]/o[0]:.1f} of {o[1]/o[0]:.1f} lanes", flush=True)
