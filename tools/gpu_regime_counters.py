"""Diagnostic, run under rocprofv3 --pmc: N contexts of one binary in one process, 1 + 3 launches of 20 frames each, in context order (dispatch order in the
counter CSV = print order here).  Prints the kernel ms per frame of every launch; the CSV carries the per-dispatch counters (translation misses, L2 read latency ...)
of a fast and a slow context side by side.   rocprofv3 --pmc ... -- python3 tools/gpu_regime_counters.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
from glrt_amd import device, host, scenes  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
frames = 20
sc, pr = scenes.CONFIGS["headline"]()
ds = []
for i in range(N):
    d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"]); d.count_rays(False)
    ds.append(d)
r = 0
for rep in range(4):
    for i, d in enumerate(ds):
        d.render_frames(pr, [host.frame_seed(frames * r + k) for k in range(frames)]); d.sync(); r += 1
        print(f"launch {r - 1}: context {i} rep {rep}: {d.stats().kernel_ms_last / frames:.4f} ms/frame", flush=True)
