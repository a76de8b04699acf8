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
    # round 6: the reference host's own tree (glrt_bvh_build_reference: one axis binned, never re-ordered) -- ties and flat boxes are where a tree shows in the image
    (22, 130, "reference", 48, 48, 8, 1, 0.0, dict(duplicates=True)),
    (23, 70, "reference", 32, 32, 6, 1, 0.0, dict(axis_aligned=True, degenerate=True)),
]




def case_scene_and_params(case):
    seed, n_tri, bvh, w, h, depth, spp, aperture, flags = case
    scene = fuzz_scene(seed, n_tri, bvh, **flags)
    far = 2.0e4 if flags.get("big_distances") else 100.0
    dist = 9000.0 if flags.get("big_distances") else 3.0
    eye = (0.0, 0.0, dist) if flags.get("axis_aligned") else (0.3 * dist, 0.2 * dist, dist)
    c2w, s2c = scenes.camera(eye, (0, 0, 0), (0, 1, 0), 45.0, w, h, 0.1, far)
    return scene, scenes.make_params(c2w, s2c, w, h, depth, spp, aperture=aperture, focal=dist)


def shadow_hostile(seed, w=48, h=32):
    """Scenes aimed at the shadow rays' closest-hit search (sampleDirect, raytrace.frag:337-403) and at what a range limit on it could get wrong
    (csrc/pt_kernel.hip.h: shadow_limit): lights lying FLUSH in the faces of the scene's bounds -- i.e. of the root box and of every ancestor's box --,
    sliver light triangles, receivers perpendicular to the lights (the ones whose light test passes) that run right up to the lights'
    edges so that the topmost light samples arrive at grazing angles (ill-conditioned triangle tests), distances of 1 .. 100 units and the whole scene
    translated to |coordinates| of up to 1e7 (an ulp of a coordinate up to 1, four orders of magnitude above EPS).  Returns (tag, scene, params)."""
    rng = np.random.default_rng(seed)
    s = float(rng.choice([1.0, 3.0, 10.0, 30.0, 100.0]))          # typical distance
    mag = float(rng.choice([0.0, 1e4, 1e5, 1e6, 1e7]))
    dirn = rng.normal(size=3)
    off = np.float32(mag) * (dirn / np.linalg.norm(dirn)).astype(np.float32)
    bvh = str(rng.choice(["sah", "lbvh", "chain"]))
    light_kind = str(rng.choice(["quad", "slivers", "fan"]))
    b = scenes.SceneBuilder()
    grey = b.add_material(scenes.diffuse(tuple(rng.uniform(0.3, 0.9, 3))))
    red = b.add_material(scenes.diffuse((0.8, 0.3, 0.3)))
    metal = b.add_material(scenes.conductor(scenes.COPPER["eta"], scenes.COPPER["kappa"], float(rng.choice([0.05, 0.3]))))
    lamp = b.add_material(scenes.emitter(tuple(rng.uniform(5.0, 20.0, 3))))
    lamp2 = b.add_material(scenes.emitter(tuple(rng.uniform(5.0, 20.0, 3))))
    top, half = 3.0 * s, 4.0 * s

    def Q(p0, e1, e2, m):
        pos, nrm = scenes.quad(tuple(p0), tuple(e1), tuple(e2))
        b.add_mesh(pos, nrm, m)

    Q((-half, 0, half), (2 * half, 0, 0), (0, 0, -2 * half), grey)                 # floor, y = 0
    Q((-half, 0, -half), (2 * half, 0, 0), (0, top, 0), red)                        # back wall, up to y = top exactly: touches the ceiling light's plane
    wx = float(rng.uniform(-0.5, 0.5)) * s
    Q((wx, 0, -half), (0, top, 0), (0, 0, 1.5 * half), metal if rng.integers(0, 2) else grey)  # a wall across the room, up to y = top: its top edge runs under the light
    # the ceiling light: in the plane y = top, the top face of the scene's bounds (nothing reaches above it), facing down
    lx, lz, lw = wx - 0.2 * s, -0.6 * half, float(rng.uniform(0.5, 2.0)) * s
    if light_kind == "quad":
        Q((lx, top, lz), (lw, 0, 0), (0, 0, lw), lamp)
    elif light_kind == "slivers":   # long thin triangles: |e1| |e2| large against the area, the triangle test's cancellation at its worst
        n = 6
        for i in range(n):
            x0 = lx + lw * i / n
            thin = lw * float(rng.choice([1e-2, 1e-3, 1e-4]))
            pos = np.array([[[x0, top, lz], [x0, top, lz + lw], [x0 + thin, top, lz + lw]]], np.float32)
            b.add_mesh(pos, np.array([[[0, -1, 0]] * 3], np.float32), lamp)
    else:                           # a fan of small triangles around a centre
        c = np.array([lx + 0.5 * lw, top, lz + 0.5 * lw])
        n = 7
        for i in range(n):
            a0, a1 = 2 * np.pi * i / n, 2 * np.pi * (i + 1) / n
            p1 = c + 0.5 * lw * np.array([np.cos(a1), 0, np.sin(a1)])
            p2 = c + 0.5 * lw * np.array([np.cos(a0), 0, np.sin(a0)])
            b.add_mesh(np.array([[c, p1, p2]], np.float32), np.array([[[0, -1, 0]] * 3], np.float32), lamp)
    # a second light flush in the +x face of the bounds, facing -x: lights the floor (perpendicular to it) and the back wall
    Q((half, 0.3 * top, -0.5 * half), (0, 0, 0.8 * half), (0, 0.5 * top, 0), lamp2)
    # a few small occluders
    pos, nrm, _ = scenes.random_triangles(int(rng.integers(0, 12)), seed + 7, 0.6 * half, 0.5 * s)
    if pos.shape[0]:
        pos[:, :, 1] = np.abs(pos[:, :, 1]) * 0.4 + 0.2 * s
        b.add_mesh(pos, nrm, grey)
    scene = b.build(bvh)
    # translate: vertices and boxes move together (float32 sums, then the tree is rebuilt from the moved vertices so that its boxes are their exact bounds)
    vert = scene["vert"].reshape(-1, 5, 3).copy()
    vert[:, 0] = (vert[:, 0] + off[None, :]).astype(np.float32)
    scene = scenes.rebuild_bvh(dict(scene, vert=vert.reshape(-1, 3)), bvh)
    eye = np.array([wx + float(rng.uniform(1.0, 3.0)) * s, float(rng.uniform(0.3, 0.95)) * top, float(rng.uniform(0.2, 0.9)) * half])
    tgt = np.array([wx, float(rng.uniform(0.6, 1.0)) * top, lz + 0.5 * lw])
    eye_w = tuple(float(v) for v in (eye.astype(np.float32) + off).astype(np.float32))
    tgt_w = tuple(float(v) for v in (tgt.astype(np.float32) + off).astype(np.float32))
    c2w, s2c = scenes.camera(eye_w, tgt_w, (0, 1, 0), float(rng.uniform(30, 70)), w, h, 0.05 * s, 100.0 * s)
    params = scenes.make_params(c2w, s2c, w, h, int(rng.integers(2, 7)), int(rng.integers(1, 4)), seed=tuple(rng.uniform(0, 1, 2)))
    return f"seed {seed}: scale {s:g}, offset {mag:g}, {light_kind} light, {bvh} tree", scene, params
