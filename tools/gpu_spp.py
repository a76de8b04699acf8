import sys; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc); d.resize(1920, 1080)
for spp in (1, 2, 4, 8):
    for depth in (8,):
        ts = []
        for f in range(5):
            d.render(dict(pr, seed=host.frame_seed(f), n_samples=spp, max_depth=depth)); d.sync(); ts.append(d.stats().kernel_ms_last)
        print(f"spp {spp} depth {depth}: {np.median(ts[1:]):.3f} ms  -> {np.median(ts[1:])/spp:.3f} ms/spp", flush=True)
for depth in (1, 2, 4):
    ts = []
    for f in range(5):
        d.render(dict(pr, seed=host.frame_seed(f), n_samples=1, max_depth=depth)); d.sync(); ts.append(d.stats().kernel_ms_last)
    print(f"spp 1 depth {depth}: {np.median(ts[1:]):.3f} ms", flush=True)
