"""Diagnostic (needs the -DGLRTX_EXPERIMENT_REALLOC build): which buffer carries a context's timing regime?  One context; before a block of launches ONE of its big buffers
(path state / queues / sample planes) is moved to a fresh allocation; 8 launches of 20 frames per block.   python tools/gpu_regimes6.py LIB [blocks]"""
import os
import pathlib
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
from glrt_amd import device, host, scenes  # noqa: E402

device.lib_path = lambda: pathlib.Path(os.path.join(ROOT, sys.argv[1]))
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 16
F = 20
sc, pr = scenes.CONFIGS["headline"]()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"]); d.count_rays(False)
names = {0: "nothing", 1: "path state", 2: "queues", 4: "sample planes"}
r = 0
seq = [0, 1, 1, 1, 2, 2, 2, 4, 4, 4, 1, 2, 4, 1, 2, 4, 0, 0][:blocks]
for b, which in enumerate(seq):
    ms = []
    for k in range(9):
        if k == 0 and which: os.environ["GLRTX_REALLOC"] = str(which)
        else: os.environ.pop("GLRTX_REALLOC", None)
        d.render_frames(pr, [host.frame_seed(F * r + i) for i in range(F)]); d.sync(); r += 1
        if k: ms.append(d.stats().kernel_ms_last / F)
    print(f"block {b:2d}: moved {names[which]:13s} median {np.median(ms):.4f}  (min {min(ms):.4f}, max {max(ms):.4f})", flush=True)
