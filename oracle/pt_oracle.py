"""ctypes front-end for oracle/_ref/libpt_oracle.so (the CPU restatement, oracle/pt_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- as the checker / the reported CPU baseline,
never as a fallback for the HIP path.
"""
from __future__ import annotations

import ctypes as C
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_LIB = _HERE / "_ref" / "libpt_oracle.so"


class _Scene(C.Structure):
    _fields_ = [("vert", C.c_void_p), ("tri", C.c_void_p), ("mat", C.c_void_p), ("light", C.c_void_p),
                ("bvh", C.c_void_p), ("n_vert", C.c_int), ("n_tri", C.c_int), ("n_mat", C.c_int),
                ("n_light", C.c_int), ("n_nodes", C.c_int)]


class _Params(C.Structure):
    _fields_ = [("c2w", C.c_float * 16), ("s2c", C.c_float * 16), ("aperture", C.c_float), ("focal", C.c_float),
                ("seed", C.c_float * 2), ("n_samples", C.c_int), ("max_depth", C.c_int), ("width", C.c_int),
                ("height", C.c_int)]


def build(force: bool = False) -> pathlib.Path:
    src = _HERE / "pt_oracle.c"
    if force or not _LIB.exists() or _LIB.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "_ref/libpt_oracle.so"], check=True, capture_output=True)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB))
        L.pt_oracle_render.restype = C.c_uint64
        L.pt_oracle_render.argtypes = [C.POINTER(_Scene), C.POINTER(_Params), C.c_void_p, C.c_size_t, C.c_int,
                                       C.c_int, C.c_int]
        L.pt_oracle_rand_stream.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                            C.c_void_p]
        L.pt_oracle_sincos.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.pt_oracle_max_threads.restype = C.c_int
        L.pt_oracle_set_ext.restype = None
        L.pt_oracle_set_ext.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.pt_oracle_resolve.restype = None
        L.pt_oracle_resolve.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


EXT_DIELECTRIC, EXT_WHITTED = 1, 2


def render(scene, params, accum=None, rows=None, threads=0, spheres=None, ext_flags=0):
    """Returns (accum (H, W, 4) float32 [L.rgb, count], n_rays).  Row 0 = bottom (gl_FragCoord.y = 0.5).
    spheres (n, 5) [cx, cy, cz, radius, material] / ext_flags: the PARITY-UNPINNED extensions (analytic spheres, EXT_DIELECTRIC,
    EXT_WHITTED) -- the CPU statement of the device kernel's extension path, nothing of the reference's."""
    L = lib()
    sph = None if spheres is None else _f32(np.asarray(spheres, np.float32).reshape(-1, 5))
    L.pt_oracle_set_ext(None if sph is None else sph.ctypes.data, 0 if sph is None else sph.shape[0], int(ext_flags))
    try:
        return _render(L, scene, params, accum, rows, threads)
    finally:
        L.pt_oracle_set_ext(None, 0, 0)


def _render(L, scene, params, accum, rows, threads):
    keep = {k: _f32(scene[k]) for k in ("vert", "tri", "mat", "light", "bvh")}
    sc = _Scene(keep["vert"].ctypes.data, keep["tri"].ctypes.data, keep["mat"].ctypes.data,
                keep["light"].ctypes.data, keep["bvh"].ctypes.data, keep["vert"].size // 15,
                keep["tri"].size // 4, keep["mat"].size // 18, keep["light"].size // 4, keep["bvh"].size // 9)
    pr = _Params()
    pr.c2w[:] = list(_f32(params["c2w"]).reshape(16))
    pr.s2c[:] = list(_f32(params["s2c"]).reshape(16))
    pr.aperture = params.get("aperture", 0.0)
    pr.focal = params.get("focal", 1.0)
    pr.seed[:] = [params["seed"][0], params["seed"][1]]
    pr.n_samples, pr.max_depth = int(params["n_samples"]), int(params["max_depth"])
    pr.width, pr.height = int(params["width"]), int(params["height"])
    h, w = pr.height, pr.width
    if accum is None:
        accum = np.zeros((h, w, 4), np.float32)
    assert accum.shape == (h, w, 4) and accum.dtype == np.float32 and accum.flags.c_contiguous
    y0, y1 = (0, h) if rows is None else rows
    n = L.pt_oracle_render(C.byref(sc), C.byref(pr), accum.ctypes.data, w * 16, y0, y1, threads)
    return accum, int(n)


def rand_stream(w, h, px, py, seed, n):
    out = np.zeros(n, np.float32)
    lib().pt_oracle_rand_stream(w, h, px, py, seed[0], seed[1], n, out.ctypes.data)
    return out


def sincos(x):
    x = _f32(x).reshape(-1)
    s = np.zeros_like(x)
    c = np.zeros_like(x)
    lib().pt_oracle_sincos(x.ctypes.data, x.size, s.ctypes.data, c.ctypes.data)
    return s, c


def resolve(accum, gamma=2.2, flip_y=False):
    """The reference's resolve pass (screen.frag:15-25 + the RGBA8 read-back of window.cpp:383-388) on an (H, W, 4) float32
    accumulator [L.rgb, count]: returns uint8 (H, W, 4).  flip_y: row 0 of the result is the top image row (saveCurrentFrame)."""
    a = np.ascontiguousarray(accum, np.float32)
    h, w = a.shape[:2]
    out = np.zeros((h, w, 4), np.uint8)
    lib().pt_oracle_resolve(a.ctypes.data, w * 16, w, h, gamma, int(flip_y), out.ctypes.data, w * 4)
    return out


def max_threads() -> int:
    return int(lib().pt_oracle_max_threads())
