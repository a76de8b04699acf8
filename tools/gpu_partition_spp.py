"""Per-rank time for world sizes 1,2,4,8 on ONE GPU (rank 0), with k samples per launch: a proxy for k frames in flight."""
import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc)
for world in (1, 2, 4, 8):
    d.set_partition(0, world, 16); d.resize(1920, 1080)
    row = []
    for spp in (1, 2, 4, 8, 16):
        ts = []
        for f in range(5):
            d.render(dict(pr, seed=host.frame_seed(f), n_samples=spp)); d.sync(); ts.append(d.stats().kernel_ms_last)
        row.append(f"{spp}spp {np.median(ts[1:])/spp:.3f}")
    print("world", world, "ms per rank per sample:", "  ".join(row), flush=True)
