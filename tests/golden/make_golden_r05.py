#!/usr/bin/env python3
"""Round 5: the shadow-hostile scenes of tests/fuzz_scenes.py (shadow_hostile: lights lying flush in the faces of their ancestors' boxes, sliver light triangles, receivers that
run up to the lights' edges so that light samples arrive at grazing angles, distances of 1 .. 100 and the scene translated to |coordinates| of up to 1e7) as golden fixtures:
same file format and the same renderer (the reference's unmodified shader on Mesa llvmpipe through oracle/glref) as make_golden.py.  They pin, on the GPU box, what the
exact shadow-ray search (the device's default since round 5) is compared with: the reference's own output, not only the oracle's.  Build container only.

    python tests/golden/make_golden_r05.py
"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "opengl-raytracer_amd" / "python"))

from fuzz_scenes import shadow_hostile  # noqa: E402
from oracle.glref import GLRef  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent
g = GLRef()
# one of every kind: light shape x tree x offset magnitude (seeds picked from the first 40 by their tags)
for seed in (0, 4, 5, 10, 14, 18, 24, 34):
    tag, scene, params = shadow_hostile(seed)
    rgb, cnt = g.render_reference(scene, params)
    name = f"shadow_hostile_seed{seed}"
    np.savez_compressed(
        OUT / f"{name}.npz",
        vert=scene["vert"], tri=scene["tri"], mat=scene["mat"], light=scene["light"], bvh=scene["bvh"],
        c2w=params["c2w"], s2c=params["s2c"],
        scalars=np.array([params["width"], params["height"], params["max_depth"], params["n_samples"]], np.int32),
        fparams=np.array([params["seed"][0], params["seed"][1], params["aperture"], params["focal"]], np.float32),
        rows=np.array((0, params["height"]), np.int32),
        frames=np.array([], np.float32).reshape(-1, 2),
        out_rgb=rgb, out_count=cnt, renderer=np.array(g.info()))
    print(f"{name}: {tag}: {rgb.shape} mean {rgb.mean():.5f} lit {np.count_nonzero(rgb.sum(-1))}")
