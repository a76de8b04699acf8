"""Diagnostic: does a context's timing regime travel with its buffers or with the stream it launches on?  C contexts x S streams created here: every context renders
on every stream (glrtx_set_stream), launches in turn; the table shows median ms per frame per (context, stream).   python tools/gpu_regimes5.py [C] [S]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
from glrt_amd import device, host, scenes  # noqa: E402

C_, S_ = (int(sys.argv[1]) if len(sys.argv) > 1 else 4), (int(sys.argv[2]) if len(sys.argv) > 2 else 5)
frames = 20
sc, pr = scenes.CONFIGS["headline"]()
ds = []
for i in range(C_):
    d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"]); d.count_rays(False)
    d.render_frames(pr, [host.frame_seed(k) for k in range(frames)]); d.sync()
    ds.append(d)
hip = ctypes.CDLL("libamdhip64.so")
streams = []
for s in range(S_):
    h = ctypes.c_void_p(); assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0; streams.append(h)
ms = np.zeros((C_, S_ + 1, 6))
r = 1
for rep in range(6):
    for si in range(S_ + 1):
        for ci, d in enumerate(ds):
            d.set_stream(0 if si == S_ else streams[si].value)   # 0: the context's own stream
            d.render_frames(pr, [host.frame_seed(frames * r + k) for k in range(frames)]); d.sync(); r += 1
            ms[ci, si, rep] = d.stats().kernel_ms_last / frames
med = np.median(ms[:, :, 1:], axis=2)
print("rows: contexts; columns: streams 0.." + str(S_ - 1) + " created here, last column: the context's own stream")
for ci in range(C_):
    print(f"context {ci}: " + "  ".join(f"{v:.4f}" for v in med[ci]))
