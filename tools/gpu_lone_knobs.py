"""Lone launches (idle device before and after) of 1 / 20 frames of the headline under run-time knobs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
from glrt_amd import device, host, scenes
sc, pr = scenes.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "headline"]()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
f0 = 0
def run(n, reps=5):
    global f0
    ts = []
    for rep in range(reps):
        seeds = [host.frame_seed(f0 + i) for i in range(n)]; f0 += n
        d.sync()
        d.render_frames(pr, seeds) if n > 1 else d.render(dict(pr, seed=seeds[0]))
        d.sync()
        st = d.stats()
        if rep: ts.append(st.kernel_ms_last)
    ts.sort()
    return ts[0], ts[len(ts)//2]
knobs = [("base", {})]
for v in (0, 8, 16, 32, 48, 64): knobs.append((f"SUSPEND_MAX={v}", {"GLRTX_SUSPEND_MAX": str(v)}))
for v in (4, 8, 24, 32, 48): knobs.append((f"REFILL_MIN={v}", {"GLRTX_REFILL_MIN": str(v)}))
for v in (0, 1024, 2048, 8192, 16384, 32768): knobs.append((f"GSS_DIV={v}", {"GLRTX_GSS_DIV": str(v)}))
knobs.append(("base again", {}))
for n in (20, 1, 48):
    print(f"--- {n} frame(s) per lone launch: kernel ms (min, median of 4)")
    for name, env in knobs:
        for k in ("GLRTX_SUSPEND_MAX", "GLRTX_REFILL_MIN", "GLRTX_GSS_DIV"): os.environ.pop(k, None)
        os.environ.update(env)
        a, b = run(n)
        print(f"{name:20s} {a:9.4f} {b:9.4f}   per frame {a/n:7.4f}", flush=True)
