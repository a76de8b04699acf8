"""Device LBVH vs the CPU SAH tree: build time and render time (16 frames in flight) on the headline scene and on config 5.
GLRTX_LIB=path selects another build of libglrtx.so (e.g. one compiled with -DGLRT_LBVH_ROTATION_PASSES=3)."""
import os, sys, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'opengl-raytracer_amd', 'python'))
import numpy as np
from glrt_amd import scenes, device, host
if os.environ.get("GLRTX_LIB"):
    device.lib_path = lambda: pathlib.Path(ROOT) / os.environ["GLRTX_LIB"]
d = device.Device()
for cfg in ("headline", "c5"):
    sc, pr = scenes.CONFIGS[cfg]()
    builds = [d.build_lbvh(sc["vert"], sc["tri"]) for _ in range(5)]
    nodes, depth, _ = builds[-1]
    res = {}
    for name, s in (("SAH", sc), ("LBVH", dict(sc, bvh=nodes, bvh_depth=depth)), ("SAH again", sc), ("LBVH again", dict(sc, bvh=nodes, bvh_depth=depth))):
        d.upload_scene(s); d.resize(pr["width"], pr["height"])
        ts = []
        for it in range(6):
            d.render_frames(pr, [host.frame_seed(1 + it * 16 + f) for f in range(16)]); d.sync(); ts.append(d.stats().kernel_ms_last / 16)
        res[name] = float(np.median(ts[1:]))
    sah = 0.5 * (res["SAH"] + res["SAH again"]); lb = 0.5 * (res["LBVH"] + res["LBVH again"])
    print(f"{os.environ.get('GLRTX_LIB', 'libglrtx.so')} {cfg}: build {min(b[2] for b in builds[1:]):.3f} ms, depth {depth}; render SAH {sah:.4f} ms/frame, device LBVH {lb:.4f} ms/frame ({(lb / sah - 1) * 100:+.2f} %)", flush=True)
