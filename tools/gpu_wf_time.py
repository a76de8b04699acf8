import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
sc, pr = scenes.CONFIGS[cfg]()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"]); d.set_variant(2)
ts = []
for f in range(8):
    d.render(dict(pr, seed=host.frame_seed(f))); d.sync(); ts.append(d.stats().kernel_ms_last)
print(os.environ.get("TAG",""), "median %.3f min %.3f" % (np.median(ts[2:]), min(ts[2:])))
