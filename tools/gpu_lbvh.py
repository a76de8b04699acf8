"""GPU LBVH: build time vs the CPU builders, and render time of config 5 (100k triangles, 1080p, 4 bounces) with either tree."""
import sys, os, time; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
d = device.Device()
for n in (100_000, 1_000_000):
    sc, pr = scenes.config_c5(n=n, bvh="chain" if n > 200_000 else "sah")
    t = time.perf_counter(); host.build_bvh(sc["vert"], sc["tri"], "sah"); t_sah = time.perf_counter() - t
    t = time.perf_counter(); cpu_nodes, cpu_depth = host.build_bvh(sc["vert"], sc["tri"], "lbvh"); t_lb = time.perf_counter() - t
    ms = []
    for i in range(4):
        t = time.perf_counter(); nodes, depth, dev_ms = d.build_lbvh(sc["vert"], sc["tri"]); wall = time.perf_counter() - t
        ms.append((dev_ms, wall * 1e3))
    print(f"{n} triangles: CPU SAH {t_sah*1e3:.0f} ms, CPU LBVH {t_lb*1e3:.0f} ms, GPU LBVH device {min(m[0] for m in ms):.2f} ms "
          f"(call incl. copies {min(m[1] for m in ms):.1f} ms), depth {depth}, equal to CPU LBVH: {np.array_equal(nodes.view(np.uint32), cpu_nodes.view(np.uint32))}", flush=True)
sc, pr = scenes.config_c5()
nodes, depth, _ = d.build_lbvh(sc["vert"], sc["tri"])
for name, s in (("SAH tree", sc), ("GPU LBVH tree", dict(sc, bvh=nodes, bvh_depth=depth))):
    d.upload_scene(s); d.resize(pr["width"], pr["height"]); d.count_rays(True); d.reset_stats()
    d.render(dict(pr, seed=host.frame_seed(0))); d.sync(); rays = d.stats().rays
    d.count_rays(False)
    ts = []
    for it in range(4):
        d.render_frames(pr, [host.frame_seed(1 + it * 8 + f) for f in range(8)]); d.sync(); ts.append(d.stats().kernel_ms_last / 8)
    print(f"config 5, {name}: stack entries {d.stats().stack_entries}, {np.median(ts[1:]):.3f} ms/frame, {rays/np.median(ts[1:])/1e3:.0f} Mrays/s ({rays} rays/frame)", flush=True)
