"""Per-rank frame time for world sizes 1,2,4,8 on ONE GPU (rank 0 of each partition): what strong scaling can give."""
import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc)
for world in (1, 2, 4, 8):
    res = []
    for rank in sorted({0, world // 2, world - 1}):
        d.set_partition(rank, world, 16); d.resize(1920, 1080)
        ts = []
        for f in range(6):
            d.render(dict(pr, seed=host.frame_seed(f))); d.sync(); ts.append(d.stats().kernel_ms_last)
        res.append((rank, round(float(np.median(ts[1:])), 3)))
    print(os.environ.get("GLRTX_BLOCK_PATHS", "auto"), "world", world, "rank times ms", res, flush=True)
