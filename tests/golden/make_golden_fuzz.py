#!/usr/bin/env python3
"""Adds the adversarial scenes of tests/fuzz_scenes.py (ties, degenerate and axis-aligned geometry, large distances,
one- and two-triangle scenes) to the golden fixtures: same file format and the same renderer (the reference's unmodified
shader on Mesa llvmpipe through oracle/glref) as make_golden.py.  Build container only.

    python tests/golden/make_golden_fuzz.py [--all]      (without --all only the cases that have no fixture yet are rendered: round 6 added seeds 22 and 23,
                                                          the reference host's own tree -- glrt_bvh_build_reference -- with duplicated and flat triangles)
"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "opengl-raytracer_amd" / "python"))

from fuzz_scenes import CASES, case_scene_and_params  # noqa: E402
from glrt_amd import host  # noqa: E402
from oracle.glref import GLRef  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent
g = GLRef()
for case in CASES:
    if "--all" not in sys.argv and (OUT / f"fuzz_seed{case[0]}.npz").exists():
        continue
    scene, params = case_scene_and_params(case)
    # one draw from cleared accumulators, as SURVEY.md 8(c) prescribes: across frames the reference re-reads its accumulator
    # through a LINEAR sampler, which at non-power-of-two sizes blends in a neighbour texel at the 1e-6 level (SURVEY F7,
    # a GL artefact outside the path)
    params = dict(params, seed=host.frame_seed(100))
    frames = []
    rgb, cnt = g.render_reference(scene, params)
    name = f"fuzz_seed{case[0]}"
    np.savez_compressed(
        OUT / f"{name}.npz",
        vert=scene["vert"], tri=scene["tri"], mat=scene["mat"], light=scene["light"], bvh=scene["bvh"],
        c2w=params["c2w"], s2c=params["s2c"],
        scalars=np.array([params["width"], params["height"], params["max_depth"], params["n_samples"]], np.int32),
        fparams=np.array([params["seed"][0], params["seed"][1], params["aperture"], params["focal"]], np.float32),
        rows=np.array((0, params["height"]), np.int32),
        frames=np.array(frames, np.float32).reshape(-1, 2),
        out_rgb=rgb, out_count=cnt, renderer=np.array(g.info()))
    print(f"{name}: {rgb.shape} mean {rgb.mean():.5f} nonzero {np.count_nonzero(rgb.sum(-1))}")
