"""Diagnostic: kernel time of a long train of identical launches in ONE process (headline, `frames` frames per launch) -- does a process sit in one timing regime
(tools/gpu_abx.py shows passes of one binary in two groups 3 % apart) or move between them?   python tools/gpu_regimes.py [frames] [launches] [gap_ms]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
from glrt_amd import device, host, scenes  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 120
gap_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
pre = sys.argv[4] if len(sys.argv) > 4 else ""   # "c": a launch of the counting kernel first, "r": read the accumulator back first, "cr": both (what tools/gpu_abx.py and bench.py do)
sc, pr = scenes.CONFIGS["headline"]()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
if "c" in pre:
    d.count_rays(True); d.render_frames(pr, [host.frame_seed(10_000 + i) for i in range(frames)]); d.sync(); d.stats()
if "r" in pre:
    d.read_accum()
d.count_rays(False)
ms = []
t0 = time.perf_counter()
for r in range(launches):
    d.render_frames(pr, [host.frame_seed(frames * r + i) for i in range(frames)])
    d.sync()
    ms.append(d.stats().kernel_ms_last / frames)
    if gap_ms > 0: time.sleep(gap_ms / 1000.0)
wall = time.perf_counter() - t0
ms = np.asarray(ms)
print(f"pre={pre!r} {launches} launches of {frames} frames, gap {gap_ms} ms, wall {wall:.2f} s: median {np.median(ms):.4f}, min {ms.min():.4f}, max {ms.max():.4f}")
print("first 12:", " ".join(f"{x:.3f}" for x in ms[:12]))
