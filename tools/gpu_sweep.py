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
knob = sys.argv[1] if len(sys.argv) > 1 else "GLRTX_REFILL_MIN"
vals = sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "2", "4", "8", "12", "16", "24", "32"]
for world in (1, 8):
    d.set_partition(0, world, 16); d.resize(1920, 1080)
    for B in (1, 16):
        out = [f"default {t(B):.3f}"]
        for v in vals:
            os.environ[knob] = v
            out.append(f"{v}: {t(B):.3f}")
        del os.environ[knob]
        print(f"world {world} B {B} {knob}:", "  ".join(out), flush=True)
