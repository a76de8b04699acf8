"""Shared generator of randomised / adversarial scenes for the fuzz tests (GPU vs oracle) and for the live check of the
oracle against the reference's shader on llvmpipe."""
import numpy as np

from glrt_amd import scenes


def fuzz_scene(seed, n_tri, bvh, duplicates=False, degenerate=False, axis_aligned=False, big_distances=False):
    rng = np.random.default_rng(seed)
    extent = 1.0 if not big_distances else 3000.0
    size = 1.1 if not big_distances else 3300.0
    pos, nrm, _ = scenes.random_triangles(n_tri, seed + 1, extent, size)
    if axis_aligned:  # boxes with zero thickness and rays that run exactly along box faces
        pos[: n_tri // 2, :, 2] = np.float32(0.25)
        pos[n_tri // 2:, :, 0] = np.float32(-0.5)
    if degenerate:  # zero-area triangles (also as lights: pdf = 1/0) and coincident vertices
        pos[::5, 2] = pos[::5, 1]
        pos[1::7, 1] = pos[1::7, 0]
    if duplicates:  # every triangle twice: exactly equal hit distances, the first one the reference visits must win
        pos = np.concatenate([pos, pos], 0)
        nrm = np.concatenate([nrm, nrm], 0)
    n = pos.shape[0]
    b = scenes.SceneBuilder()
    mats = [b.add_material(scenes.diffuse(tuple(rng.uniform(0.0, 1.0, 3)))) for _ in range(5)]
    mats += [b.add_material(scenes.conductor(scenes.COPPER["eta"], scenes.COPPER["kappa"], float(a))) for a in (0.02, 0.3, 1.0)]
    mats += [b.add_material(scenes.emitter(tuple(rng.uniform(0.0, 30.0, 3)))) for _ in range(2)]
    mats += [b.add_material(scenes.media()), b.add_material(scenes.diffuse((0.0, 0.0, 0.0)))]
    b.add_mesh(pos, nrm, np.asarray(mats)[rng.integers(0, len(mats), n)])
    return b.build(bvh)


CASES = [
    # seed, triangles, bvh, width, height, depth, spp, aperture, flags
    (11, 60, "sah", 37, 29, 8, 2, 0.0, {}),
    (12, 200, "lbvh", 64, 40, 6, 1, 0.05, {}),
    (13, 80, "chain", 33, 17, 5, 2, 0.0, {}),
    (14, 120, "sah", 48, 48, 8, 1, 0.0, dict(duplicates=True)),
    (15, 90, "lbvh", 40, 40, 6, 2, 0.0, dict(duplicates=True, degenerate=True)),
    (16, 150, "sah", 50, 31, 7, 1, 0.0, dict(degenerate=True)),
    (17, 64, "sah", 32, 32, 6, 1, 0.0, dict(axis_aligned=True)),
    (18, 100, "sah", 40, 24, 5, 1, 0.0, dict(big_distances=True)),
    (19, 1, "sah", 16, 16, 4, 2, 0.0, {}),
    (20, 2, "chain", 16, 16, 4, 1, 0.0, dict(duplicates=True)),
    (21, 300, "sah", 96, 54, 16, 1, 0.2, {}),
]




def case_scene_and_params(case):
    seed, n_tri, bvh, w, h, depth, spp, aperture, flags = case
    scene = fuzz_scene(seed, n_tri, bvh, **flags)
    far = 2.0e4 if flags.get("big_distances") else 100.0
    dist = 9000.0 if flags.get("big_distances") else 3.0
    eye = (0.0, 0.0, dist) if flags.get("axis_aligned") else (0.3 * dist, 0.2 * dist, dist)
    c2w, s2c = scenes.camera(eye, (0, 0, 0), (0, 1, 0), 45.0, w, h, 0.1, far)
    return scene, scenes.make_params(c2w, s2c, w, h, depth, spp, aperture=aperture, focal=dist)
