"""All BASELINE configs at full size on one GPU: rays/frame, ms/frame with 1 and 8 frames per launch."""
import sys, os, time; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
d = device.Device()
for name, kw in (("c1", {}), ("c2", {}), ("c3", {}), ("c3", dict(bvh="sah")), ("c4", dict(n_samples=1)), ("c5", {}), ("headline", {})):
    sc, pr = scenes.CONFIGS[name](**kw)
    d.upload_scene(sc); d.set_partition(0, 1, 16); d.resize(pr["width"], pr["height"])
    d.count_rays(True); d.reset_stats(); d.render(dict(pr, seed=host.frame_seed(0))); d.sync(); rays = d.stats().rays; d.count_rays(False)
    res = []
    for B in (1, 8):
        ts = []
        for it in range(4):
            d.render_frames(pr, [host.frame_seed(1 + it * B + f) for f in range(B)]); d.sync(); ts.append(d.stats().kernel_ms_last / B)
        res.append(float(np.median(ts[1:])))
    print(f"{name} {kw} {pr['width']}x{pr['height']} depth {pr['max_depth']} tris {sc['tri'].shape[0]} bvh {sc['bvh_kind']}: {rays/1e6:.2f} Mrays/frame, "
          f"{res[0]:.3f} ms/frame ({rays/res[0]/1e3:.0f} Mrays/s) one launch per frame, {res[1]:.3f} ms/frame ({rays/res[1]/1e3:.0f} Mrays/s) 8 in flight", flush=True)
