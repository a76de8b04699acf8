"""In-process sweep of launch knobs (environment variables read at every launch) with frames in flight."""
import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc)
def t(B, reps=4):
    ts = []
    for it in range(reps):
        d.render_frames(pr, [host.frame_seed(it * B + f) for f in range(B)]); d.sync(); ts.append(d.stats().kernel_ms_last / B)
    return float(np.median(ts[1:]))
for world in (1, 8):
    d.set_partition(0, world, 16); d.resize(1920, 1080)
    for B in (8, 16, 32):
        base = t(B)
        out = [f"base {base:.3f}"]
        for k, vals in (("GLRTX_BLOCK_PATHS", ("1024", "2048")), ("GLRTX_GSS_DIV", ("0",))):
            for v in vals:
                os.environ[k] = v
                out.append(f"{k[6:]}={v} {t(B):.3f}")
            del os.environ[k]
        print(f"world {world} B {B}:", "  ".join(out), flush=True)
