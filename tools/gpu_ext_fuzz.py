"""Diagnostic: the extension kernel (analytic spheres, dielectric, Whitted termination; parity unpinned) against its CPU statement on hostile sphere lists --
radii of 0, negative, 1e-30, 1e30, inf, NaN, centres at infinity, overlapping and nested spheres, up to 40 of them -- and indices of refraction of 0, 1, 1e-30,
1e30, negative, NaN, with every combination of the two flags.   python tools/gpu_ext_fuzz.py SEED N"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
import numpy as np
from glrt_amd import device, scenes
from glrt_amd.scenes import SceneBuilder, quad, conductor, diffuse, emitter, dielectric, camera, make_params
from oracle import pt_oracle
rng = np.random.default_rng(int(sys.argv[1])); N = int(sys.argv[2])
spec = [0.0, -0.0, -1.0, 1e-30, 1e30, np.inf, -np.inf, np.nan, 1e-45, 3e38, 1.0]
d = device.Device(); W, H = 48, 36
bad = refused = 0
for it in range(N):
    b = SceneBuilder()
    ior = float(rng.choice([1.5, 1.0, 0.0, 1e-30, 1e30, -1.5, np.nan, np.inf, 0.7, 2.4]))
    mats = [b.add_material(diffuse((0.7, 0.7, 0.7))), b.add_material(diffuse((0.8, 0.3, 0.3))), b.add_material(dielectric(ior)),
            b.add_material(conductor(scenes.COPPER["eta"], scenes.COPPER["kappa"], 0.2))]
    lamp = b.add_material(emitter((10.0, 10.0, 10.0)))
    b.add_mesh(*quad((-10, 0, 10), (20, 0, 0), (0, 0, -20)), mats[0])
    b.add_mesh(*quad((-1, 5, -1), (2, 0, 0), (0, 0, 2)), lamp)
    if rng.integers(0, 3) == 0: b.add_mesh(*scenes.icosphere(1, 0.8, (0.0, 0.8, 2.0)), mats[2])  # a glass icosphere on the triangle path
    n = int(rng.integers(0, 41))
    sph = np.zeros((n, 5), np.float32)
    for i in range(n):
        sph[i, :3] = rng.uniform(-3, 3, 3) + [0, 1.5, 0]
        sph[i, 3] = rng.uniform(0.05, 1.2)
        sph[i, 4] = mats[int(rng.integers(0, 4))]
        if rng.integers(0, 5) == 0: sph[i, int(rng.integers(0, 4))] = spec[int(rng.integers(0, len(spec)))]
    sc = b.build("sah")
    c2w, s2c = camera((0, 3, 9), (0, 1, 0), (0, 1, 0), 40.0, W, H)
    p = make_params(c2w, s2c, W, H, int(rng.integers(1, 9)), int(rng.integers(1, 3)), seed=(float(rng.uniform()), float(rng.uniform())))
    flags = int(rng.integers(0, 4))
    d.upload_scene(sc)
    try:
        d.upload_spheres(sph if n else None)
    except device.GlrtxError as e:
        assert e.code == device.GLRTX_ESCENE, e
        refused += 1
        continue
    ref, rays = pt_oracle.render(sc, p, spheres=sph if n else None, ext_flags=flags)
    d.set_extensions(flags); d.set_partition(0, 1, 16); d.resize(W, H); d.clear(); d.reset_stats(); d.count_rays(True)
    d.render(p); d.sync()
    acc = d.read_accum()
    same = (acc.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(acc) & np.isnan(ref))
    if not same.all() or d.stats().rays != rays:
        bad += 1
        print(f"MISMATCH it {it}: {int((~same).any(-1).sum())} pixels, rays {d.stats().rays} / {rays}, flags {flags}, ior {ior}, n {n}, depth {p['max_depth']}", flush=True)
        np.save(os.path.join(ROOT, "gpurun_out", f"ext_fuzz_bad_{sys.argv[1]}_{it}.npy"), sph)
d.set_extensions(0); d.upload_spheres(None)
print(f"done: {N} scenes, {refused} sphere lists refused, mismatches {bad}")
