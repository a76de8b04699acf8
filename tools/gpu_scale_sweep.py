"""Diagnostic: BASELINE config 1's scene scaled by powers of ten (geometry, BVH boxes and camera alike) -- where does the device still equal the oracle?
Small scales drive products into the denormal range (flushed on both sides), large ones into overflow.   python tools/gpu_scale_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
import numpy as np
from glrt_amd import device, scenes
from oracle import pt_oracle
d = device.Device()
W, H = int(os.environ.get("SWEEP_W", "48")), int(os.environ.get("SWEEP_H", "32"))
CFG = os.environ.get("SWEEP_CFG", "c1")
for kind in ("sah", "chain"):
    sc0, pr0 = scenes.config_c1(W, H, max_depth=4, n_samples=1, bvh=kind, subdiv=1) if CFG == "c1" else scenes.config_c2(W, H, 4, 1, kind, 1)
    for k in (1e-18, 1e-12, 1e-9, 1e-6, 1e-3, 1e3, 1e6, 1e9, 1e12, 1e15, 1e18):
        kf = np.float32(k)
        vert = sc0["vert"].reshape(-1, 5, 3).copy(); vert[:, 0] *= kf
        nodes = sc0["bvh"].reshape(-1, 9).copy(); nodes[:, 0:6] *= kf
        sc = dict(sc0, vert=vert.reshape(-1, 3), bvh=nodes.reshape(-1, 3))
        c2w = np.array(pr0["c2w"], np.float32).reshape(4, 4).copy(); c2w[3, :3] *= kf
        p = dict(pr0, c2w=c2w.reshape(-1), focal=float(pr0.get("focal", 1.0)))
        ref, rays = pt_oracle.render(sc, p)
        d.upload_scene(sc); d.resize(W, H); d.clear(); d.count_rays(True); d.reset_stats(); d.render(p); d.sync()
        acc = d.read_accum()
        diff = int((acc.view(np.uint32) != ref.view(np.uint32)).any(-1).sum())
        lit = float((ref[..., :3].sum(-1) > 0).mean())
        print(f"{CFG} {kind} scale {k:g}: {diff} of {acc.shape[0] * acc.shape[1]} pixels differ, rays {d.stats().rays} / {rays}, lit {lit:.2f}, finite {bool(np.isfinite(ref).all())}", flush=True)
