"""Per-rank frame time for world sizes 1,2,4,8 on ONE GPU (rank 0 of each partition), per kernel variant."""
import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc)
for world in (1, 2, 4, 8):
    out = []
    for v in (0, 1, 2):
        d.set_variant(v); d.set_partition(0, world, 16); d.resize(1920, 1080)
        ts = []
        for f in range(6):
            d.render(dict(pr, seed=host.frame_seed(f))); d.sync(); ts.append(d.stats().kernel_ms_last)
        out.append(round(float(np.median(ts[1:])), 3))
    print("world", world, "ms per rank: tile", out[0], "persistent", out[1], "wgwf", out[2], flush=True)
