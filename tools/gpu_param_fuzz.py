"""Diagnostic: hostile render parameters -- camera matrices with NaN / inf / zero entries, aperture and focal length NaN / inf / negative / zero, zero samples,
zero depth -- device against oracle (NaNs as NaNs).   python tools/gpu_param_fuzz.py SEED N"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
import numpy as np
from glrt_amd import device, scenes
from oracle import pt_oracle
rng = np.random.default_rng(int(sys.argv[1])); N = int(sys.argv[2])
specials = [np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-30, 1e30, -1.0, 1e-45, 3e38]
sc, pr0 = scenes.config_c1(40, 28, max_depth=3, n_samples=2, subdiv=1)
d = device.Device(); d.upload_scene(sc); d.resize(40, 28)
bad = refused = 0
for it in range(N):
    p = dict(pr0)
    c2w = np.array(p["c2w"], np.float32).copy(); s2c = np.array(p["s2c"], np.float32).copy()
    for _ in range(int(rng.integers(0, 3))):
        m = c2w if rng.integers(0, 2) else s2c
        m[int(rng.integers(0, 16))] = specials[int(rng.integers(0, len(specials)))] if rng.integers(0, 2) else float(rng.normal()) * 10
    p["c2w"], p["s2c"] = c2w, s2c
    if rng.integers(0, 3) == 0: p["aperture"] = float(specials[int(rng.integers(0, len(specials)))])
    if rng.integers(0, 3) == 0: p["focal"] = float(specials[int(rng.integers(0, len(specials)))])
    if rng.integers(0, 5) == 0: p["n_samples"] = int(rng.integers(0, 4))
    if rng.integers(0, 5) == 0: p["max_depth"] = int(rng.integers(0, 4))
    if rng.integers(0, 4) == 0: p["seed"] = (float(specials[int(rng.integers(0, len(specials)))]), float(rng.uniform()))
    try:
        d.clear(); d.count_rays(True); d.reset_stats(); d.render(p); d.sync()
    except device.GlrtxError as e:
        refused += 1; continue
    ref, rays = pt_oracle.render(sc, p)
    acc = d.read_accum()
    same = (acc.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(acc) & np.isnan(ref))
    if not same.all() or d.stats().rays != rays:
        bad += 1
        print(f"MISMATCH it {it}: {int((~same).any(-1).sum())} pixels, rays {d.stats().rays} / {rays}, params aperture {p.get('aperture')} focal {p.get('focal')} ns {p['n_samples']} depth {p['max_depth']} seed {p['seed']}\n  c2w {c2w.tolist()}\n  s2c {s2c.tolist()}", flush=True)
print(f"done: {N} parameter sets, refused {refused}, mismatches {bad}")
