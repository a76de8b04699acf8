"""Build container only: the random-parameter scenes of tests/test_gpu_fuzz.py::test_fuzz_random_parameters (triangle count, tree builder, image size, depth,
samples, lens, flags drawn from the seed) through the reference's own shader on llvmpipe against the oracle, one frame each (a second accumulated frame at
a non-power-of-two size meets the reference's GL_LINEAR sampler).  The GPU test compares the device with the oracle on these seeds; this
compares the oracle with the reference on them.   python tools/live_soak.py FIRST_SEED N"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_scenes import fuzz_scene  # noqa: E402
from glrt_amd import host, scenes  # noqa: E402
from oracle import glref, pt_oracle  # noqa: E402

if not glref.reference_available():
    raise SystemExit("reference checkout / Mesa llvmpipe not present")
gl = glref.GLRef()
first, n = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    flags = {k: bool(rng.integers(0, 4) == 0) for k in ("duplicates", "degenerate", "axis_aligned")}
    n_tri = int(rng.integers(1, 400))
    bvh = ("sah", "lbvh", "chain")[int(rng.integers(0, 3))] if n_tri < 120 else ("sah", "lbvh")[int(rng.integers(0, 2))]
    bvh = os.environ.get("GLRT_FUZZ_BVH", bvh)  # (as tests/test_gpu_fuzz.py: a soak under one builder)
    w, h = int(rng.integers(1, 70)), int(rng.integers(1, 50))
    depth, spp = int(rng.integers(0, 17)), int(rng.integers(1, 4))
    aperture = float(rng.choice([0.0, 0.0, 0.1]))
    scene = fuzz_scene(seed, n_tri, bvh, **flags)
    eye = tuple(float(v) for v in rng.uniform(-3.5, 3.5, 3))
    if flags["axis_aligned"]:
        eye = (0.0, 0.0, 3.0)
    c2w, s2c = scenes.camera(eye, (0, 0, 0), (0, 1, 0), float(rng.uniform(20, 90)), w, h, 0.1, 100.0)
    params = scenes.make_params(c2w, s2c, w, h, depth, spp, aperture=aperture, focal=3.0)
    p = dict(params, seed=host.frame_seed(int(rng.integers(0, 10_000))))
    rgb, cnt = gl.render_reference(scene, p)
    acc, _ = pt_oracle.render(scene, p)
    same = np.array_equal(acc[..., :3].view(np.uint32), np.asarray(rgb, np.float32).view(np.uint32)) and np.array_equal(acc[..., 3].view(np.uint32), np.asarray(cnt, np.float32).view(np.uint32))
    if not same:
        bad += 1
        nd = int((acc[..., :3].view(np.uint32) != np.asarray(rgb, np.float32).view(np.uint32)).any(-1).sum())
        print(f"seed {seed}: {n_tri} tris {bvh} {w}x{h} depth {depth} spp {spp} {flags}: {nd} pixels differ", flush=True)
    if (seed - first) % 25 == 24:
        print(f"... {seed - first + 1} seeds, {bad} with differences", flush=True)
print(f"{n} seeds from {first}: {bad} with differences between oracle and live reference")
