"""Time one prebuilt library variant (lib name via GLRTX_LIB) on a config; checks bit-equality vs the tile kernel."""
import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
if os.environ.get("GLRTX_LIB"):
    device.lib_path = lambda: device.LIB_DIR / os.environ["GLRTX_LIB"]
cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
v = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sc, pr = scenes.CONFIGS[cfg]()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
d.set_variant(0); d.render(dict(pr, seed=host.frame_seed(0))); d.sync(); ref = d.read_accum()
d.set_variant(v); d.clear(); d.render(dict(pr, seed=host.frame_seed(0))); d.sync(); img = d.read_accum()
ts = []
for f in range(8):
    d.render(dict(pr, seed=host.frame_seed(f + 1))); d.sync(); ts.append(d.stats().kernel_ms_last)
print(os.environ.get("GLRTX_LIB", "default"), cfg, "v%d" % v, "identical", np.array_equal(ref.view(np.uint32), img.view(np.uint32)), "median %.3f min %.3f" % (np.median(ts[2:]), min(ts[2:])), flush=True)
