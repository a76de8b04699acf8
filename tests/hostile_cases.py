"""Hostile inputs of round 4's fuzzes as (tag, scene, params) cases: determinants beyond 2^126 (denormal reciprocals), scenes scaled by powers of ten, seeds far
outside [0, 1), extreme material parameters, non-finite vertices.  tests/test_reference_live.py runs them through the live reference (oracle == llvmpipe, build
container only); tests/test_gpu_parity.py holds the device-against-oracle forms of the same inputs."""
import numpy as np

from glrt_amd import scenes
from glrt_amd.scenes import SceneBuilder, camera, conductor, diffuse, emitter, make_params, quad


def cases(w=48, h=32):
    inf, nan = float("inf"), float("nan")
    # determinants of 4e37 .. 1.7e38: triangles with edges of ~1e19 and one corner in front of the camera
    b = SceneBuilder()
    grey, lamp = b.add_material(diffuse((0.7, 0.6, 0.5))), b.add_material(emitter((8.0, 8.0, 8.0)))
    E, pos, rng = 1.3e19, [], np.random.default_rng(5)
    for i in range(9):
        v0 = np.array([-0.6 + 0.1 * i, -0.5 + 0.07 * i, -2.0 - 0.2 * i])
        k = rng.uniform(0.5, 1.0, 2)
        pos.append([v0, v0 + [E * k[0], 0.0, -0.1 * E * (i % 3)], v0 + [0.0, E * k[1], 0.05 * E * (i % 2)]])
    b.add_mesh(np.array(pos), np.array([[[0, 0, 1]] * 3] * 9), grey)
    b.add_mesh(np.array([[[-1.5, 1.0, -1.0], [-1.0, 1.0, -1.0], [-1.5, 1.0, -1.6]], [[-1.0, 1.0, -1.0], [-1.0, 1.0, -1.6], [-1.5, 1.0, -1.6]]]),
               np.array([[[0, -1, 0]] * 3] * 2), lamp)
    c2w, s2c = camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 60.0, w, h, 0.1, 100.0)
    for kind in ("chain", "sah"):
        yield f"huge determinants, {kind} tree", b.build(kind), make_params(c2w, s2c, w, h, 3, 2)
    # scaled scenes
    for kind in ("sah", "chain"):
        sc0, pr0 = scenes.config_c1(w, h, max_depth=4, n_samples=1, bvh=kind, subdiv=1)
        for k in (1e-12, 1e-3, 1e3, 1e4, 1e6, 1e12):
            kf = np.float32(k)
            vert = sc0["vert"].reshape(-1, 5, 3).copy()
            vert[:, 0] *= kf
            nodes = sc0["bvh"].reshape(-1, 9).copy()
            nodes[:, 0:6] *= kf
            c = np.array(pr0["c2w"], np.float32).reshape(4, 4).copy()
            c[3, :3] *= kf
            yield f"scale {k:g}, {kind} tree", dict(sc0, vert=vert.reshape(-1, 3), bvh=nodes.reshape(-1, 3)), dict(pr0, c2w=c.reshape(-1))
    # seeds
    sc, pr = scenes.config_c1(w, h, max_depth=4, n_samples=2, subdiv=1)
    for seed in [(-0.5, 1.5), (1234.5, -77.25), (2.0e7, 0.5), (3.0e9, 0.5), (1.0e20, -1.0e20), (inf, 0.5), (nan, 0.5)]:
        yield f"seed {seed}", sc, dict(pr, seed=seed)
    # materials
    g = diffuse((0.7, 0.7, 0.7))
    mats = [(f"conductor alpha {a:g}", [g, g, conductor((0.2, 0.9, 1.1), (3.9, 2.4, 2.2), a), conductor((1.5,) * 3, (0.0,) * 3, a)], (10.0,) * 3) for a in (1e-8, 1e-2, 1e4)]
    mats += [(f"conductor eta {e:g} kappa {k:g}", [g, conductor((e,) * 3, (k,) * 3, 0.1), conductor((e,) * 3, (k,) * 3, 0.5), g], (10.0,) * 3)
             for e, k in ((0.0, 0.0), (1e3, 1e-3), (1e19, 1e19), (-1.0, 2.0))]
    mats += [(f"albedo {alb}", [diffuse(alb), g, diffuse(alb), g], (10.0,) * 3) for alb in ((0.0,) * 3, (10.0, 5.0, 1.0), (-1.0, 0.5, 2.0), (1e30, 1e-30, 1e-40), (inf, 0.5, 0.5), (nan, 0.5, 0.5))]
    mats += [(f"emitter {e}", [g] * 4, e) for e in ((1e30,) * 3, (1e-30, 1e-40, 0.0), (-5.0, 1.0, 1.0), (inf, 1.0, 1.0), (nan, 1.0, 1.0))]
    for tag, ms, lamp_e in mats:
        b = SceneBuilder()
        ids = [b.add_material(m) for m in ms]
        lamp = b.add_material(emitter(lamp_e))
        b.add_mesh(*quad((-4, 0, 4), (8, 0, 0), (0, 0, -8)), ids[0])
        b.add_mesh(*quad((-4, 0, -4), (8, 0, 0), (0, 6, 0)), ids[1])
        b.add_mesh(*quad((-2.5, 0.01, 1.0), (2, 0, 0), (0, 2, -1)), ids[2])
        b.add_mesh(*quad((0.5, 0.01, 1.0), (2, 0, 0), (0, 2, -1)), ids[3])
        b.add_mesh(*quad((-1.5, 5.5, -1.5), (3, 0, 0), (0, 0, 3)), lamp)
        c2w, s2c = camera((0, 2.5, 8), (0, 1.0, 0), (0, 1, 0), 45.0, w, h)
        yield tag, b.build("sah"), make_params(c2w, s2c, w, h, 6, 2, seed=(0.31, 0.62))
    # non-finite vertices
    sc0, pr0 = scenes.config_c1(w, h, max_depth=4, n_samples=1, bvh="sah", subdiv=1)
    for tag, val in (("nan", np.nan), ("+inf", np.inf), ("-inf", -np.inf), ("3e38", 3e38), ("mixed", None)):
        vert = sc0["vert"].reshape(-1, 5, 3).copy()
        if val is None:
            vert[7, 0, 1], vert[8, 0, 1], vert[100, 0, 0], vert[101, 0, 0], vert[333, 0, 2] = np.inf, -np.inf, 3e38, -3e38, np.nan
        else:
            vert[7, 0, 1] = vert[100, 0, 0] = vert[333, 0, 2] = val
        for kind in ("sah", "chain", "lbvh"):
            yield f"vertices {tag}, {kind} tree", scenes.rebuild_bvh(dict(sc0, vert=vert.reshape(-1, 3)), kind), pr0
