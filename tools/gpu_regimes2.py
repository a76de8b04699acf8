"""Diagnostic: is the 3 % timing regime of a process (tools/gpu_regimes.py, tools/gpu_abx.py) a property of where its buffers landed?  One process creates the
context N times -- each time behind a dummy hipMalloc of another size, so that the context's buffers land elsewhere -- and times 8 launches of 20 frames
each time.   python tools/gpu_regimes2.py [N]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
from glrt_amd import device, host, scenes  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
frames = 20
sc, pr = scenes.CONFIGS["headline"]()
hip = ctypes.CDLL("libamdhip64.so")
pads = [0, 1 << 20, 3 << 20, 64 << 20, (64 << 20) + (1 << 16), 1 << 30, (1 << 30) + (2 << 20), 5 << 20, 0, 0, 17 << 20, 129 << 20]
for it in range(N):
    pad = pads[it % len(pads)]
    p = ctypes.c_void_p()
    if pad: assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(pad)) == 0
    d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"]); d.count_rays(False)
    ms = []
    for r in range(9):
        d.render_frames(pr, [host.frame_seed(frames * r + i) for i in range(frames)]); d.sync()
        ms.append(d.stats().kernel_ms_last / frames)
    d.close()
    if pad: hip.hipFree(p)
    ms = np.asarray(ms[1:])
    print(f"context {it}: dummy allocation {pad:>11d} B in front: median {np.median(ms):.4f} min {ms.min():.4f} max {ms.max():.4f}", flush=True)
