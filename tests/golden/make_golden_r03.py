#!/usr/bin/env python3
"""Round-3 additions to tests/golden/: the reference's UNMODIFIED fragment shader on Mesa llvmpipe (see make_golden.py for the rules --
a fixture is inputs and outputs only) on scenes that exercise what round 3 changed in the device code:

  comb63            a comb BVH that keeps one stack entry per level, 62 deep -- the reference's `int stack[64]` (raytrace.frag:284) at its
                    limit, the per-lane LDS stack of the hand-written step
  one_child_forks   forks that leave out children.x or children.y (raytrace.frag:299-307): the never-hit record, the stack budget
  c2_small_3frames  the headline scene at 128x64, 8 bounces, three accumulated frames: leaf pairs (two chained triangle records per fork with
                    two leaf children), parked rays, overlapped single-frame launches
  tris2000_lbvh     2000 random triangles under the linear BVH (deep, unbalanced: BASELINE config 5's kind of tree)
  c4_small_spp4     BASELINE config 4's scene at subdivision 2, four samples per pixel in one pass

Run in the build container only:  make -C oracle && make -C opengl-raytracer_amd host && python tests/golden/make_golden_r03.py
"""
from __future__ import annotations

import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "opengl-raytracer_amd" / "python"))

from glrt_amd import host, scenes  # noqa: E402
from oracle.glref import GLRef  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent
g = GLRef()


def save(name, scene, params, frames=None):
    rgb, cnt = g.render_reference(scene, params, frames=frames)
    np.savez_compressed(
        OUT / f"{name}.npz",
        vert=scene["vert"], tri=scene["tri"], mat=scene["mat"], light=scene["light"], bvh=scene["bvh"],
        c2w=params["c2w"], s2c=params["s2c"],
        scalars=np.array([params["width"], params["height"], params["max_depth"], params["n_samples"]], np.int32),
        fparams=np.array([params["seed"][0], params["seed"][1], params["aperture"], params["focal"]], np.float32),
        rows=np.array((0, params["height"]), np.int32),
        frames=np.array(frames if frames is not None else np.zeros((0, 2)), np.float32).reshape(-1, 2),
        out_rgb=rgb, out_count=cnt, renderer=np.array(g.info()))
    print(f"{name}: {rgb.shape} mean {rgb.mean():.5f} nonzero {np.count_nonzero(rgb.sum(-1))}")


# comb of 63 triangles: the chain BVH with children.x / children.y swapped, so that every fork pushes a leaf first and the rest of the tree second
scene, params = scenes._random_tri_scene(63, 20260103, 0.9, 4.5, 64, 48, 3, 1, "chain")  # (63 triangles close together: most pixels see one)
nodes = scene["bvh"].reshape(-1, 9).copy()
fk = nodes[:, 8] < 0
nodes[fk, 6], nodes[fk, 7] = nodes[fk, 7].copy(), nodes[fk, 6].copy()
save("comb63", dict(scene, bvh=nodes.reshape(-1, 3)), params)

# forks with one child left out (the scene of test_forks_with_an_absent_child_match_the_oracle)
scene, params = scenes.config_c1(96, 64, max_depth=5, n_samples=2, subdiv=1)
nodes = scene["bvh"].reshape(-1, 9).copy()
rng = np.random.default_rng(5)
forks = np.flatnonzero(nodes[:, 8] < 0)
extra = []
for k, f in enumerate(rng.choice(forks, 40, replace=False)):
    side = 6 + (k & 1)
    child = int(nodes[f, side])
    unary = nodes[child].copy()
    unary[6:9] = (-1.0, -1.0, -1.0)
    unary[6 + ((k >> 1) & 1)] = float(child)
    nodes[f, side] = float(nodes.shape[0] + len(extra))
    extra.append(unary)
save("one_child_forks", dict(scene, bvh=np.concatenate([nodes, np.array(extra, np.float32)], 0).reshape(-1, 3)), params)

# the headline scene, small, three frames (a power-of-two size: the reference reads the previous frame through a LINEAR sampler at
# uv = fragCoord / windowSize, which lands exactly on texel centres only then -- SURVEY.md F7; at 160x90 its own frames bleed by an ulp)
scene, params = scenes.config_headline(width=128, height=64)
save("c2_small_3frames", scene, params, frames=[host.frame_seed(f) for f in range(3)])

# 2000 random triangles under the linear BVH (the CPU statement of the device builder: Morton order, rotations, small subtrees rebuilt) --
# BASELINE config 5's kind of tree, deep and unbalanced, many leaf pairs
scene, params = scenes._random_tri_scene(2000, 20260104, 1.6, 6.0, 96, 64, 4, 1, "lbvh")
save("tris2000_lbvh", scene, dict(params, seed=host.frame_seed(7)))

# BASELINE config 4's scene (eight spheres on a ground quad under a lamp, no box) at subdivision 2, 4 samples per pixel in one pass, 8 bounces
scene, params = scenes.config_c4(width=96, height=54, n_samples=4, subdiv=2)
save("c4_small_spp4", scene, dict(params, seed=host.frame_seed(11)))
