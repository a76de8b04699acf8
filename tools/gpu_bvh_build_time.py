"""Device time of the two GPU tree builders (glrtx_build_lbvh, glrtx_build_bvh_sah) against the triangle count, best of 5 builds each (the first builds allocate).
    python tools/gpu_bvh_build_time.py [n ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
import numpy as np
from glrt_amd import device, scenes
d = device.Device()
for n in [int(a) for a in sys.argv[1:]] or [1000, 10_000, 100_000, 1_000_000]:
    ext = 10.0 * (n / 100_000.0) ** (1.0 / 3.0)
    pos, nrm, _ = scenes.random_triangles(n, 20260102, ext, 0.15)
    vert = np.zeros((n * 3, 5, 3), np.float32); vert[:, 0] = pos.reshape(-1, 3); vert[:, 1] = nrm.reshape(-1, 3)
    tri = np.concatenate([np.arange(n * 3, dtype=np.float32).reshape(n, 3), np.zeros((n, 1), np.float32)], 1)
    out = []
    for name, fn in (("LBVH (Morton + 4 rotation sweeps + 64-leaf rebuilds)", d.build_lbvh), ("SAH by levels (+ 64-leaf exact sweep)", d.build_bvh_sah)):
        t = time.perf_counter()
        ms = [fn(vert.reshape(-1, 3), tri)[1:] for _ in range(6)]
        wall = (time.perf_counter() - t) / 6 * 1e3
        out.append(f"{name}: device {min(m[1] for m in ms[1:]):7.3f} ms (call incl. copies {wall:7.2f} ms), depth {ms[-1][0]}")
    print(f"{n:8d} triangles: " + "; ".join(out), flush=True)
