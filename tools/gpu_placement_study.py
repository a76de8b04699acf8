"""Diagnostic (needs the -DGLRTX_EXPERIMENT_PLACEMENT build): M buffers of the path-state size from hipMalloc, all alive; for each the streaming-write bandwidth and the time of a
read-modify-write kernel in the renderer's pattern (glrtx_debug_probe_placement), then a context whose path state lives in that buffer (GLRTX_STATE_PTR) and its kernel ms per frame,
launches in turn.  Which probe predicts the render time?   python tools/gpu_placement_study.py LIB [M]"""
import ctypes as C
import os
import pathlib
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
from glrt_amd import device, host, scenes  # noqa: E402

device.lib_path = lambda: pathlib.Path(os.path.join(ROOT, sys.argv[1]))
M = int(sys.argv[2]) if len(sys.argv) > 2 else 16
F = 20
BYTES = 6 * 41472000 * 16
sc, pr = scenes.CONFIGS["headline"]()
hip = C.CDLL("libamdhip64.so")
L = device.lib()
L.glrtx_debug_probe_placement.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_double)]
bufs = []
for i in range(M):
    p = C.c_void_p()
    if hip.hipMalloc(C.byref(p), C.c_size_t(BYTES)) != 0: break
    hip.hipMemset(p, 0, C.c_size_t(BYTES)); bufs.append(p)
hip.hipDeviceSynchronize()
probe = []
for rep in range(2):
    row = []
    for p in bufs:
        out = (C.c_double * 2)(); assert L.glrtx_debug_probe_placement(p, BYTES, out) == 0
        row.append((out[0], out[1]))
    probe.append(row)
ds = []
for p in bufs:
    os.environ["GLRTX_STATE_PTR"] = hex(p.value)
    d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"]); d.count_rays(False)
    d.render_frames(pr, [host.frame_seed(k) for k in range(F)]); d.sync()
    ds.append(d)
os.environ.pop("GLRTX_STATE_PTR")
ms = [[] for _ in ds]
r = 1
for rep in range(6):
    for i, d in enumerate(ds):
        d.render_frames(pr, [host.frame_seed(F * r + k) for k in range(F)]); d.sync(); r += 1
        ms[i].append(d.stats().kernel_ms_last / F)
med = np.asarray([np.median(m) for m in ms])
w = np.asarray([[x[0] for x in row] for row in probe]); t = np.asarray([[x[1] for x in row] for row in probe])
print("buffer            address   write GB/s (two passes)   rmw ms (two passes)   render ms/frame")
for i, p in enumerate(bufs):
    print(f"{i:3d}  {p.value:#018x}   {w[0, i]:7.0f} {w[1, i]:7.0f}          {t[0, i]:.3f} {t[1, i]:.3f}         {med[i]:.4f}", flush=True)
print(f"correlation with render ms/frame: write bandwidth {np.corrcoef(med, w.mean(0))[0, 1]:+.3f}, rmw time {np.corrcoef(med, t.mean(0))[0, 1]:+.3f}")
