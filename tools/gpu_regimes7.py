"""Diagnostic (needs the -DGLRTX_EXPERIMENT_STATE_BW build): does a context's render time follow the streaming-WRITE bandwidth its path-state buffer happens to have?
N contexts in one process: median kernel ms per frame (launches in turn) against glrtx_debug_state_write_bw.   python tools/gpu_regimes7.py LIB [N]"""
import ctypes as C
import os
import pathlib
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
from glrt_amd import device, host, scenes  # noqa: E402

device.lib_path = lambda: pathlib.Path(os.path.join(ROOT, sys.argv[1]))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
F = 20
sc, pr = scenes.CONFIGS["headline"]()
ds = []
envs = [e for e in sys.argv[3:]]  # optional: one "K=V,K=V" environment per context (cycled), set while the context allocates
for i in range(N):
    kv = dict(x.split("=", 1) for x in envs[i % len(envs)].split(",") if x) if envs else {}
    for k in ("GLRTX_STATE_ARENA_GB", "GLRTX_STATE_ARENA_OFF_GB"): os.environ.pop(k, None)
    os.environ.update(kv)
    d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"]); d.count_rays(False)
    d.render_frames(pr, [host.frame_seed(k) for k in range(F)]); d.sync()
    ds.append(d)
for k in ("GLRTX_STATE_ARENA_GB", "GLRTX_STATE_ARENA_OFF_GB"): os.environ.pop(k, None)
L = device.lib()
L.glrtx_debug_state_write_bw.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
ms = [[] for _ in ds]
r = 1
for rep in range(8):
    for i, d in enumerate(ds):
        d.render_frames(pr, [host.frame_seed(F * r + k) for k in range(F)]); d.sync(); r += 1
        ms[i].append(d.stats().kernel_ms_last / F)
rows = []
for i, d in enumerate(ds):
    g = C.c_double(); assert L.glrtx_debug_state_write_bw(d.h, C.byref(g)) == 0
    rows.append((np.median(ms[i]), g.value))
    tag = envs[i % len(envs)] if envs else ""
    print(f"context {i} [{tag}]: {rows[-1][0]:.4f} ms/frame   path-state write bandwidth {rows[-1][1]:.0f} GB/s", flush=True)
a = np.asarray(rows)
print(f"correlation of ms/frame with write bandwidth: {np.corrcoef(a[:, 0], a[:, 1])[0, 1]:+.3f}")
