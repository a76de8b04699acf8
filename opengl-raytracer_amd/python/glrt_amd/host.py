"""ctypes binding of libglrt_host.so (include/glrt_host.h): BVH builders and camera matrices.

CPU-only; loadable without a GPU.  Raises if the library has not been built
(`make -C opengl-raytracer_amd host` or `__graft_entry__.build()`).
"""
from __future__ import annotations

import ctypes as C
import pathlib

import numpy as np

PKG_ROOT = pathlib.Path(__file__).resolve().parents[2]
LIB_DIR = PKG_ROOT / "lib"

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = LIB_DIR / "libglrt_host.so"
        if not path.exists():
            raise RuntimeError(f"{path} is missing: run `make -C {PKG_ROOT}` (or __graft_entry__.build())")
        L = C.CDLL(str(path))
        fp = C.POINTER(C.c_float)
        L.glrt_bvh_node_count.restype = C.c_size_t
        L.glrt_bvh_node_count.argtypes = [C.c_size_t]
        L.glrt_bvh_build_sah.argtypes = [fp, C.c_size_t, fp, C.c_size_t, fp, C.POINTER(C.c_int)]
        L.glrt_bvh_build_lbvh.argtypes = [fp, C.c_size_t, fp, C.c_size_t, fp, C.POINTER(C.c_int)]
        L.glrt_bvh_build_sah_levels.argtypes = [fp, C.c_size_t, fp, C.c_size_t, fp, C.POINTER(C.c_int)]
        L.glrt_bvh_build_chain.argtypes = [fp, C.c_size_t, fp, C.c_size_t, fp]
        L.glrt_bvh_build_reference.argtypes = [fp, C.c_size_t, fp, C.c_size_t, fp, C.POINTER(C.c_int)]
        L.glrt_bvh_lights_first.argtypes = [fp, C.c_size_t, fp, C.c_size_t, fp, C.c_size_t]
        L.glrt_bvh_order_by_hits.argtypes = [fp, C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t]
        L.glrt_bvh_add_shadow_hits.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, fp, fp, C.c_size_t]
        L.glrt_bvh_reinsert.argtypes = [fp, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.glrt_look_at.argtypes = [fp, fp, fp, fp]
        L.glrt_perspective.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, fp]
        L.glrt_mat4_mul.argtypes = [fp, fp, fp]
        L.glrt_mat4_inverse.argtypes = [fp, fp]
        L.glrt_frame_seed.argtypes = [C.c_uint32, fp]
        _lib = L
    return _lib


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def build_bvh(vert: np.ndarray, tri: np.ndarray, kind: str = "sah"):
    """vert: (nV*5, 3) float32 texels, tri: (nT, 4).  Returns (nodes (nN*3, 3) float32, max_depth)."""
    L = lib()
    vert = _f32(vert).reshape(-1, 15)
    tri = _f32(tri).reshape(-1, 4)
    n = int(L.glrt_bvh_node_count(tri.shape[0]))
    nodes = np.zeros((n * 3, 3), np.float32)
    depth = C.c_int(0)
    if kind == "sah-reinsert":  # "sah" + the insertion-based optimisation pass (as glrt::Scene::parse's builder of that name)
        nodes, depth = build_bvh(vert, tri, "sah")
        out, d2, _, _ = reinsert(nodes)
        return out, (d2 if d2 >= 0 else depth)
    if kind == "sah":
        rc = L.glrt_bvh_build_sah(_fp(vert), vert.shape[0], _fp(tri), tri.shape[0], _fp(nodes), C.byref(depth))
    elif kind == "lbvh":
        rc = L.glrt_bvh_build_lbvh(_fp(vert), vert.shape[0], _fp(tri), tri.shape[0], _fp(nodes), C.byref(depth))
    elif kind == "sahl":  # binned SAH by levels + exact sweep at the bottom: the CPU statement of the device builder glrtx_build_bvh_sah
        rc = L.glrt_bvh_build_sah_levels(_fp(vert), vert.shape[0], _fp(tri), tri.shape[0], _fp(nodes), C.byref(depth))
    elif kind == "reference":  # the reference host's own tree, restated rule for rule (bvh.cpp:72-160); never re-ordered afterwards
        rc = L.glrt_bvh_build_reference(_fp(vert), vert.shape[0], _fp(tri), tri.shape[0], _fp(nodes), C.byref(depth))
    elif kind == "chain":
        rc = L.glrt_bvh_build_chain(_fp(vert), vert.shape[0], _fp(tri), tri.shape[0], _fp(nodes))
        depth.value = 2
    else:
        raise ValueError(kind)
    if rc != 0:
        raise RuntimeError(f"glrt_bvh_build_{kind} failed: {rc}")
    return nodes, depth.value


def lights_first(nodes, tri, mat):
    """glrt_bvh_lights_first on a copy of `nodes`: the child whose subtree holds the emitting triangles into the slot the traversal visits first, at every fork where
    only one child has any.  Returns (nodes (nN*3, 3) float32, forks exchanged).  GLRT_BVH_LIGHTS_FIRST=0 in the environment returns the tree unchanged."""
    import os
    out = _f32(nodes).reshape(-1, 3).copy()
    if os.environ.get("GLRT_BVH_LIGHTS_FIRST", "1") == "0":
        return out, 0
    tri, mat = _f32(tri).reshape(-1, 4), _f32(mat).reshape(-1, 18)
    rc = lib().glrt_bvh_lights_first(_fp(out), out.shape[0] // 3, _fp(tri), tri.shape[0], _fp(mat), mat.shape[0])
    if rc < 0:
        raise RuntimeError(f"glrt_bvh_lights_first failed: {rc}")
    return out, int(rc)


def order_by_hits(nodes, tri_hits, tri=None, mat=None):
    """glrt_bvh_order_by_hits on a copy of `nodes`: at every fork the child whose subtree collected more closest hits in a calibration frame (device.Device.hit_histogram)
    into the slot the traversal visits first.  With `tri` and `mat` the shadow rays' share is added first (glrt_bvh_add_shadow_hits).  Returns (nodes, forks exchanged)."""
    out = _f32(nodes).reshape(-1, 3).copy()
    h = np.ascontiguousarray(tri_hits, dtype=np.uint32).copy()
    if tri is not None and mat is not None:
        t, m = _f32(tri).reshape(-1, 4), _f32(mat).reshape(-1, 18)
        rc = lib().glrt_bvh_add_shadow_hits(h.ctypes.data_as(C.POINTER(C.c_uint32)), h.shape[0], _fp(t), _fp(m), m.shape[0])
        if rc < 0:
            raise RuntimeError(f"glrt_bvh_add_shadow_hits failed: {rc}")
    rc = lib().glrt_bvh_order_by_hits(_fp(out), out.shape[0] // 3, h.ctypes.data_as(C.POINTER(C.c_uint32)), h.shape[0])
    if rc < 0:
        raise RuntimeError(f"glrt_bvh_order_by_hits failed: {rc}")
    return out, int(rc)


def reinsert(nodes, max_passes: int = 8):
    """glrt_bvh_reinsert on a copy of `nodes`: insertion-based optimisation of a finished tree.  Returns (nodes (nN*3, 3) float32, max depth (-1: left alone),
    subtrees moved, (cost before, cost after)) -- cost = summed area of the forks' boxes / the root's."""
    out = _f32(nodes).reshape(-1, 3).copy()
    depth = C.c_int(0)
    cost = (C.c_double * 2)()
    rc = lib().glrt_bvh_reinsert(_fp(out), out.shape[0] // 3, int(max_passes), C.byref(depth), cost)
    if rc == -3:  # GLRT_HOST_EDEPTH: the optimised tree would be deeper than the traversal stack allows; `out` is the input tree again
        return out, -1, 0, (cost[0], cost[1])
    if rc < 0:
        raise RuntimeError(f"glrt_bvh_reinsert failed: {rc}")
    return out, depth.value, int(rc), (cost[0], cost[1])


def look_at(eye, center, up) -> np.ndarray:
    out = np.zeros(16, np.float32)
    lib().glrt_look_at(_fp(_f32(eye)), _fp(_f32(center)), _fp(_f32(up)), _fp(out))
    return out


def perspective(fovy_deg, aspect, z_near, z_far) -> np.ndarray:
    out = np.zeros(16, np.float32)
    lib().glrt_perspective(fovy_deg, aspect, z_near, z_far, _fp(out))
    return out


def mat4_inverse(m) -> np.ndarray:
    out = np.zeros(16, np.float32)
    if lib().glrt_mat4_inverse(_fp(_f32(m)), _fp(out)) != 0:
        raise RuntimeError("singular matrix")
    return out


def mat4_mul(a, b) -> np.ndarray:
    out = np.zeros(16, np.float32)
    lib().glrt_mat4_mul(_fp(_f32(a)), _fp(_f32(b)), _fp(out))
    return out


def frame_seed(frame: int):
    out = np.zeros(2, np.float32)
    lib().glrt_frame_seed(frame, _fp(out))
    return float(out[0]), float(out[1])
