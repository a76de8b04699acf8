"""Diagnostic: does the timing regime of a context (tools/gpu_ab_inproc.py: contexts of one binary differ by up to 3 %) follow where its SCENE arrays landed?  One
context; the scene is uploaded again and again (nodes, normals, materials, lights re-allocated; path state, queues and planes stay where they are), a dummy
allocation in between so that the arrays land elsewhere; 10 launches of 20 frames after each upload.   python tools/gpu_regimes3.py [uploads]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
from glrt_amd import device, host, scenes  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
frames = 20
sc, pr = scenes.CONFIGS["headline"]()
hip = ctypes.CDLL("libamdhip64.so")
d = device.Device(); d.resize(pr["width"], pr["height"]); d.count_rays(False)
keep = []
r = 0
for it in range(N):
    d.upload_scene(sc)
    ms = []
    for _ in range(11):
        d.render_frames(pr, [host.frame_seed(frames * r + i) for i in range(frames)]); d.sync(); r += 1
        ms.append(d.stats().kernel_ms_last / frames)
    ms = np.asarray(ms[1:])
    print(f"upload {it}: median {np.median(ms):.4f} min {ms.min():.4f} max {ms.max():.4f}", flush=True)
    p = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t((1 + it % 5) << 20)); keep.append(p)
