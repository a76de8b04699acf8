"""Synthetic scenes in the reference's flat buffer wire format.

The reference ships no scenes (its .gitignore excludes scenes/*), so the
BASELINE.json configurations are generated here, deterministically, as the five
buffers Scene::parse uploads (scene.cpp:254-269):

  vert  (nV*5, 3)  pos, normal, uv, tangent, binormal     trimesh.h:15-25
  tri   (nT, 4)    i, j, k, materialId  (as floats)        scene.h:16-18
  mat   (nM*6, 3)  type, emission, param0, param1, param2, texIds   scene.h:28-35
  light (nL, 4)    copy of every emissive triangle         scene.cpp:246-248
  bvh   (nN*3, 3)  bboxMin, bboxMax, children              bvh.h:84-100

Like the reference's OBJ path, meshes carry three fresh vertices per triangle
(trimesh.cpp:186-187).  "Spheres" are icospheres: the reference has no analytic
sphere primitive.
"""
from __future__ import annotations

import numpy as np

from . import host

MTRL_EMITTER, MTRL_DIFFUSE, MTRL_CONDUCTOR, MTRL_DIELECTRIC, MTRL_MEDIA = 1, 2, 3, 4, 5


# --------------------------------------------------------------------------- materials
def emitter(emission):
    return dict(type=MTRL_EMITTER, emission=emission)


def diffuse(reflectance):
    return dict(type=MTRL_DIFFUSE, param0=reflectance)


def conductor(eta, kappa, alpha):
    # scene.cpp:152-162: param0 = kappa, param1 = eta, param2 = (alpha, alpha, alpha)
    return dict(type=MTRL_CONDUCTOR, param0=kappa, param1=eta, param2=(alpha, alpha, alpha))


def media():
    return dict(type=MTRL_MEDIA)


def _mat_rows(m):
    z = (0.0, 0.0, 0.0)
    t = float(m["type"])
    return [(t, t, t), m.get("emission", z), m.get("param0", z), m.get("param1", z), m.get("param2", z), z]


# --------------------------------------------------------------------------- meshes
def _icosahedron():
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], np.int64)
    return v, f


def icosphere(subdiv: int, radius: float, center):
    """Returns (pos (nT,3,3), nrm (nT,3,3)) float32; 20*4**subdiv triangles, smooth unit normals."""
    v, f = _icosahedron()
    tris = v[f]  # (20,3,3)
    for _ in range(subdiv):
        a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
        ab, bc, ca = a + b, b + c, c + a
        ab /= np.linalg.norm(ab, axis=1, keepdims=True)
        bc /= np.linalg.norm(bc, axis=1, keepdims=True)
        ca /= np.linalg.norm(ca, axis=1, keepdims=True)
        tris = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1),
                               np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)], 0)
    nrm = tris.astype(np.float32)
    pos = (tris * radius + np.asarray(center, np.float64)).astype(np.float32)
    return pos, nrm


def quad(p0, e1, e2):
    """Two triangles p0, p0+e1, p0+e1+e2, p0+e2 with flat normal normalize(cross(e1, e2))."""
    p0, e1, e2 = (np.asarray(x, np.float64) for x in (p0, e1, e2))
    n = np.cross(e1, e2)
    n /= np.linalg.norm(n)
    a, b, c, d = p0, p0 + e1, p0 + e1 + e2, p0 + e2
    pos = np.array([[a, b, c], [a, c, d]], np.float32)
    nrm = np.broadcast_to(n.astype(np.float32), pos.shape).copy()
    return pos, nrm


def random_triangles(n: int, seed: int, extent: float, size: float = 0.15):
    """centres U(-extent, extent)^3, vertices centre + size*U(-1,1)^3; flat normals."""
    rng = np.random.default_rng(seed)
    ctr = rng.uniform(-extent, extent, (n, 1, 3))
    pos = (ctr + size * rng.uniform(-1.0, 1.0, (n, 3, 3))).astype(np.float32)
    e1 = pos[:, 1].astype(np.float64) - pos[:, 0]
    e2 = pos[:, 2].astype(np.float64) - pos[:, 0]
    nn = np.cross(e1, e2)
    ln = np.linalg.norm(nn, axis=1, keepdims=True)
    nn = np.where(ln > 0, nn / np.maximum(ln, 1e-30), np.array([0.0, 0.0, 1.0]))
    nrm = np.repeat(nn[:, None, :], 3, 1).astype(np.float32)
    return pos, nrm, rng


# --------------------------------------------------------------------------- assembly
class SceneBuilder:
    def __init__(self):
        self.materials = []
        self._pos, self._nrm, self._mid = [], [], []

    def add_material(self, m) -> int:
        self.materials.append(m)
        return len(self.materials) - 1

    def add_mesh(self, pos, nrm, material_id):
        pos = np.asarray(pos, np.float32).reshape(-1, 3, 3)
        nrm = np.asarray(nrm, np.float32).reshape(-1, 3, 3)
        mid = np.broadcast_to(np.asarray(material_id, np.float32), (pos.shape[0],))
        self._pos.append(pos)
        self._nrm.append(nrm)
        self._mid.append(mid.copy())

    def build(self, bvh: str = "sah"):
        pos = np.concatenate(self._pos, 0)
        nrm = np.concatenate(self._nrm, 0)
        mid = np.concatenate(self._mid, 0)
        n_tri = pos.shape[0]
        vert = np.zeros((n_tri * 3, 5, 3), np.float32)
        vert[:, 0] = pos.reshape(-1, 3)
        vert[:, 1] = nrm.reshape(-1, 3)
        vert = vert.reshape(-1, 3)
        idx = np.arange(n_tri * 3, dtype=np.float32).reshape(n_tri, 3)
        tri = np.concatenate([idx, mid[:, None]], 1).astype(np.float32)
        mat = np.array([r for m in self.materials for r in _mat_rows(m)], np.float32)
        emissive = np.array([float(np.linalg.norm(np.asarray(m.get("emission", (0, 0, 0)), np.float64))) != 0.0
                             for m in self.materials])
        light = tri[emissive[mid.astype(np.int64)]]
        built, depth = host.build_bvh(vert, tri, bvh)
        # the light side first (host.lights_first: glrt_bvh_lights_first, as glrt::Scene::parse applies it); `bvh_builder` keeps the builder's own output
        # (the reference host's own tree -- "reference" -- is never re-ordered: its point is the reference's own visiting order)
        nodes, swapped = host.lights_first(built, tri, mat) if bvh not in ("chain", "reference") else (built, 0)
        return dict(vert=vert, tri=tri, mat=mat, light=np.ascontiguousarray(light.reshape(-1, 4)), bvh=nodes,
                    bvh_depth=depth, bvh_kind=bvh, bvh_builder=built, bvh_lights_first=swapped)


def rebuild_bvh(scene, kind: str):
    s = dict(scene)
    s["bvh_builder"], s["bvh_depth"] = host.build_bvh(scene["vert"], scene["tri"], kind)
    s["bvh"], s["bvh_lights_first"] = host.lights_first(s["bvh_builder"], scene["tri"], scene["mat"]) if kind not in ("chain", "reference") else (s["bvh_builder"], 0)
    s["bvh_kind"] = kind
    return s


def camera(origin, target, up, fov_deg, width, height, near=0.1, far=100.0):
    """(c2w, s2c) as the reference computes them: window.cpp:230-233 with modelM = I (scene.cpp:93,113)."""
    view = host.look_at(origin, target, up)
    proj = host.perspective(float(fov_deg), float(width) / float(height), near, far)
    return host.mat4_inverse(view), host.mat4_inverse(proj)


def make_params(c2w, s2c, width, height, max_depth, n_samples=1, seed=(0.137, 0.731), aperture=0.0, focal=1.0):
    return dict(c2w=np.asarray(c2w, np.float32), s2c=np.asarray(s2c, np.float32), width=int(width),
                height=int(height), max_depth=int(max_depth), n_samples=int(n_samples),
                seed=(float(np.float32(seed[0])), float(np.float32(seed[1]))), aperture=float(aperture),
                focal=float(focal))


COPPER = dict(eta=(0.200, 0.924, 1.102), kappa=(3.912, 2.452, 2.142))


# --------------------------------------------------------------------------- BASELINE configs
def config_c1(width=256, height=256, max_depth=1, n_samples=1, bvh="sah", subdiv=2):
    """3 icospheres + ground quad + emitter quad; 964 triangles at subdiv 2."""
    b = SceneBuilder()
    grey = b.add_material(diffuse((0.7, 0.7, 0.7)))
    red = b.add_material(diffuse((0.8, 0.3, 0.3)))
    cu = b.add_material(conductor(COPPER["eta"], COPPER["kappa"], 0.2))
    lamp = b.add_material(emitter((10.0, 10.0, 10.0)))
    b.add_mesh(*quad((-10, 0, 10), (20, 0, 0), (0, 0, -20)), grey)  # normal +y
    b.add_mesh(*icosphere(subdiv, 1.0, (-2.2, 1.0, 0.0)), red)
    b.add_mesh(*icosphere(subdiv, 1.0, (0.0, 1.0, 0.0)), cu)
    b.add_mesh(*icosphere(subdiv, 1.0, (2.2, 1.0, 0.0)), grey)
    b.add_mesh(*quad((-1, 5, -1), (2, 0, 0), (0, 0, 2)), lamp)  # normal -y
    scene = b.build(bvh)
    c2w, s2c = camera((0, 3, 9), (0, 1, 0), (0, 1, 0), 40.0, width, height)
    return scene, make_params(c2w, s2c, width, height, max_depth, n_samples)


def dielectric(ior, tint=(1.0, 1.0, 1.0)):
    """EXTENSION material (no reference counterpart: MTRL_DIELECTRIC = 4 is declared at raytrace.frag:32 and never branched on, so
    without GLRTX_EXT_DIELECTRIC it renders black, like in the reference): param0 = tint, param1.x = index of refraction."""
    return dict(type=4, param0=tint, param1=(float(ior), 0.0, 0.0))


def config_spheres(width=256, height=256, max_depth=4, n_samples=1, subdiv=None, glass=False, bvh="sah"):
    """BASELINE configs[0] read literally -- "3 spheres + 1 ground plane": the C1 layout with the three spheres ANALYTIC
    (subdiv None; returned as a third value, (n, 5) rows [cx, cy, cz, radius, material] for glrtx_upload_spheres) or, for the
    tessellation-limit comparison, as icospheres of the given subdivision on the pinned triangle path (third value None).
    glass: the middle sphere is a dielectric (ior 1.5) instead of copper.  PARITY UNPINNED (no spheres in the reference)."""
    b = SceneBuilder()
    grey = b.add_material(diffuse((0.7, 0.7, 0.7)))
    red = b.add_material(diffuse((0.8, 0.3, 0.3)))
    mid = b.add_material(dielectric(1.5) if glass else conductor(COPPER["eta"], COPPER["kappa"], 0.2))
    lamp = b.add_material(emitter((10.0, 10.0, 10.0)))
    b.add_mesh(*quad((-10, 0, 10), (20, 0, 0), (0, 0, -20)), grey)
    b.add_mesh(*quad((-1, 5, -1), (2, 0, 0), (0, 0, 2)), lamp)
    balls = [((-2.2, 1.0, 0.0), red), ((0.0, 1.0, 0.0), mid), ((2.2, 1.0, 0.0), grey)]
    spheres = None
    if subdiv is None:
        spheres = np.array([[c[0], c[1], c[2], 1.0, m] for c, m in balls], np.float32)
    else:
        for c, m in balls:
            b.add_mesh(*icosphere(subdiv, 1.0, c), m)
    scene = b.build(bvh)
    c2w, s2c = camera((0, 3, 9), (0, 1, 0), (0, 1, 0), 40.0, width, height)
    return scene, make_params(c2w, s2c, width, height, max_depth, n_samples), spheres


def _eight_spheres(b: SceneBuilder, subdiv: int):
    xs = (-3.3, -1.1, 1.1, 3.3)
    albedos = ((0.75, 0.75, 0.75), (0.8, 0.3, 0.3), (0.3, 0.7, 0.35), (0.3, 0.4, 0.8))
    for x, a in zip(xs, albedos):
        b.add_mesh(*icosphere(subdiv, 1.0, (x, 1.0, 1.5)), b.add_material(diffuse(a)))
    for x, alpha in zip(xs, (0.05, 0.1, 0.2, 0.4)):
        b.add_mesh(*icosphere(subdiv, 1.0, (x, 3.0, -1.8)),
                   b.add_material(conductor(COPPER["eta"], COPPER["kappa"], alpha)))


def config_c2(width=1920, height=1080, max_depth=4, n_samples=1, bvh="sah", subdiv=3):
    """Cornell-style box (5 quads) + 8 icospheres + ceiling emitter; 10,252 triangles at subdiv 3."""
    b = SceneBuilder()
    white = b.add_material(diffuse((0.75, 0.75, 0.75)))
    red = b.add_material(diffuse((0.75, 0.25, 0.25)))
    green = b.add_material(diffuse((0.25, 0.75, 0.25)))
    lamp = b.add_material(emitter((15.0, 15.0, 15.0)))
    b.add_mesh(*quad((-5, 0, 5), (10, 0, 0), (0, 0, -10)), white)    # floor, +y
    b.add_mesh(*quad((-5, 10, -5), (10, 0, 0), (0, 0, 10)), white)   # ceiling, -y
    b.add_mesh(*quad((-5, 0, -5), (10, 0, 0), (0, 10, 0)), white)    # back, +z
    b.add_mesh(*quad((-5, 0, 5), (0, 0, -10), (0, 10, 0)), red)      # left, +x
    b.add_mesh(*quad((5, 0, -5), (0, 0, 10), (0, 10, 0)), green)     # right, -x
    _eight_spheres(b, subdiv)
    b.add_mesh(*quad((-2, 9.99, -2), (4, 0, 0), (0, 0, 4)), lamp)    # -y
    scene = b.build(bvh)
    c2w, s2c = camera((0, 5, 16), (0, 4.2, 0), (0, 1, 0), 40.0, width, height)
    return scene, make_params(c2w, s2c, width, height, max_depth, n_samples)


def config_headline(width=1920, height=1080, n_samples=1, bvh="sah", subdiv=3):
    """BASELINE.json metric: the C2 scene at 1920x1080, 8 bounces (u_maxDepth = 8)."""
    return config_c2(width, height, 8, n_samples, bvh, subdiv)


def _random_tri_scene(n, seed, extent, cam_z, width, height, max_depth, n_samples, bvh):
    pos, nrm, rng = random_triangles(n, seed, extent)
    b = SceneBuilder()
    palette = [b.add_material(diffuse(tuple(rng.uniform(0.2, 0.9, 3)))) for _ in range(64)]
    lamp = b.add_material(emitter((2.0, 2.0, 2.0)))
    is_light = rng.uniform(0.0, 1.0, n) < 0.2
    mid = np.where(is_light, lamp, np.asarray(palette)[rng.integers(0, 64, n)])
    b.add_mesh(pos, nrm, mid)
    scene = b.build(bvh)
    c2w, s2c = camera((0, 0, cam_z), (0, 0, 0), (0, 1, 0), 40.0, width, height, 0.1, 200.0)
    return scene, make_params(c2w, s2c, width, height, max_depth, n_samples)


def config_c3(width=1920, height=1080, max_depth=1, n_samples=1, bvh="chain", n=10_000):
    """10k random triangles; bvh='chain' is the brute-force linear scan, 'sah' the same scene with a real tree."""
    return _random_tri_scene(n, 20260101, 4.0, 14.0, width, height, max_depth, n_samples, bvh)


def config_c4(width=3840, height=2160, max_depth=8, n_samples=16, bvh="sah", subdiv=3):
    """8 icospheres + ground + emitter (no box)."""
    b = SceneBuilder()
    grey = b.add_material(diffuse((0.7, 0.7, 0.7)))
    lamp = b.add_material(emitter((12.0, 12.0, 12.0)))
    b.add_mesh(*quad((-15, 0, 15), (30, 0, 0), (0, 0, -30)), grey)
    _eight_spheres(b, subdiv)
    b.add_mesh(*quad((-3, 9.0, -3), (6, 0, 0), (0, 0, 6)), lamp)
    scene = b.build(bvh)
    c2w, s2c = camera((0, 5, 16), (0, 2.0, 0), (0, 1, 0), 40.0, width, height)
    return scene, make_params(c2w, s2c, width, height, max_depth, n_samples)


def config_c5(width=1920, height=1080, max_depth=4, n_samples=1, bvh="sah", n=100_000):
    return _random_tri_scene(n, 20260102, 10.0, 34.0, width, height, max_depth, n_samples, bvh)


CONFIGS = {"c1": config_c1, "c2": config_c2, "c3": config_c3, "c4": config_c4, "c5": config_c5,
           "headline": config_headline}


def scene_bytes(scene) -> int:
    """Algorithmic bytes of one read of the compact scene: 72 B/triangle (3 pos + 3 normals),
    16 B/triangle (indices + material), 36 B/node, 72 B/material, 16 B/light."""
    n_tri = scene["tri"].reshape(-1, 4).shape[0]
    n_node = scene["bvh"].reshape(-1, 9).shape[0]
    n_mat = scene["mat"].reshape(-1, 18).shape[0]
    n_light = scene["light"].reshape(-1, 4).shape[0]
    return 88 * n_tri + 36 * n_node + 72 * n_mat + 16 * n_light


# --------------------------------------------------------------------------- JSON + OBJ export (Scene::parse input)
def export_json_obj(builder: SceneBuilder, directory, width, height, origin, target, up, fov_deg, near=0.1, far=100.0,
                    aperture=None, focal=None, name="scene.json"):
    """Write `builder`'s meshes as one OBJ per add_mesh() call plus the JSON scene description the
    reference's Scene::parse reads (schema: SURVEY.md Appendix C).  Floats are written with 9 significant
    digits, which round-trips float32 exactly.  Returns the JSON path."""
    import json
    import pathlib

    d = pathlib.Path(directory)
    d.mkdir(parents=True, exist_ok=True)
    shapes = []
    for k, (pos, nrm, mid) in enumerate(zip(builder._pos, builder._nrm, builder._mid)):
        assert np.all(mid == mid[0]), "one material per shape (scene.cpp:124-219)"
        m = builder.materials[int(mid[0])]
        lines = []
        for p in pos.reshape(-1, 3):
            lines.append("v %.9g %.9g %.9g" % tuple(p))
        for n in nrm.reshape(-1, 3):
            lines.append("vn %.9g %.9g %.9g" % tuple(n))
        for t in range(pos.shape[0]):
            a = 3 * t + 1
            lines.append(f"f {a}//{a} {a + 1}//{a + 1} {a + 2}//{a + 2}")
        (d / f"shape{k}.obj").write_text("\n".join(lines) + "\n")
        sh = {"type": "obj", "filename": f"shape{k}.obj"}
        if m["type"] == MTRL_DIFFUSE:
            sh.update(material="diffuse", reflectance=[float(v) for v in m["param0"]])
        elif m["type"] == MTRL_CONDUCTOR:
            sh.update(material="conductor", kappa=[float(v) for v in m["param0"]], eta=[float(v) for v in m["param1"]],
                      alpha=float(m["param2"][0]))
        elif m["type"] == MTRL_EMITTER:
            sh.update(material="emitter", emission=[float(v) for v in m["emission"]])
        else:
            raise ValueError("export supports diffuse / conductor / emitter shapes")
        shapes.append(sh)
    cam = {"type": "perspective", "fov": fov_deg, "nearClip": near, "farClip": far,
           "lookAt": {"origin": list(origin), "target": list(target), "up": list(up)}}
    if aperture is not None:
        cam["apertureRadius"] = aperture
    if focal is not None:
        cam["focalLength"] = focal
    doc = {"film": {"width": width, "height": height}, "camera": cam, "scene": shapes}
    (d / name).write_text(json.dumps(doc, indent=1))
    return d / name
