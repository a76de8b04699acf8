"""Diagnostic: wire-format scenes with a few floats mutated (tree children and boxes, triangle indices and materials, lights, material parameters, vertices:
NaN, +-inf, -1, 0.5, 2^24+1, 3e38, copies of other entries, ...).  What glrtx_upload_scene accepts must render like the oracle; what it refuses must be refused
with GLRTX_ESCENE / GLRTX_EDEPTH.   python tools/gpu_wire_fuzz.py SEED N"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
import numpy as np
from glrt_amd import device, scenes
from oracle import pt_oracle
rng = np.random.default_rng(int(sys.argv[1])); N = int(sys.argv[2])
specials = np.array([np.nan, np.inf, -np.inf, -1.0, -0.0, 0.0, 0.5, 1.0, 2.0, 1e9, 1.7e7, 16777216.0, 16777217.0, 3e38, -3e38, 2147483648.0, 4294967296.0, 1e-40], np.float32)
bases = []
for kind in ("sah", "chain", "lbvh"):
    bases.append(scenes.config_c3(32, 24, n=37, bvh=kind, max_depth=3))
    bases.append(scenes.config_c1(32, 24, bvh=kind, subdiv=1, max_depth=3))
d = device.Device()
acc_n = ref_n = bad = 0
for it in range(N):
    sc0, pr = bases[it % len(bases)]
    sc = dict(sc0)
    key = ("bvh", "bvh", "bvh", "tri", "light", "mat", "vert")[int(rng.integers(0, 7))]
    a = np.array(sc[key], np.float32).copy().reshape(-1)
    for _ in range(int(rng.integers(1, 6))):
        i = int(rng.integers(0, a.size)); mode = int(rng.integers(0, 4))
        if mode == 0: a[i] = specials[int(rng.integers(0, specials.size))]
        elif mode == 1: a[i] = a[int(rng.integers(0, a.size))]
        elif mode == 2: a[i] = float(rng.integers(-5, a.size))
        else: a[i] = a[i] + 1.0
    sc[key] = a.reshape(np.shape(sc[key]))
    try:
        d.upload_scene(sc)
    except device.GlrtxError as e:
        assert e.code in (device.GLRTX_ESCENE, device.GLRTX_EDEPTH), e
        ref_n += 1
        continue
    acc_n += 1
    ref, rays = pt_oracle.render(sc, pr)
    d.resize(32, 24); d.clear(); d.count_rays(True); d.reset_stats(); d.render(pr); d.sync()
    acc = d.read_accum()
    same = (acc.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(acc) & np.isnan(ref))
    if not same.all() or d.stats().rays != rays:
        bad += 1
        print(f"MISMATCH it {it} key {key}: {int((~same).any(-1).sum())} pixels, rays {d.stats().rays} / {rays}", flush=True)
        np.savez(os.path.join(ROOT, "gpurun_out", f"wire_fuzz_bad_{sys.argv[1]}_{it}.npz"), **{k: np.asarray(v) for k, v in sc.items() if not isinstance(v, (str, int))})
    if it % 100 == 99: print(f"{it + 1}: accepted {acc_n}, refused {ref_n}, mismatches {bad}", flush=True)
print(f"done: accepted {acc_n}, refused {ref_n}, mismatches {bad}")
