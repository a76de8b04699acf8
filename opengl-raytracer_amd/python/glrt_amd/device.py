"""ctypes binding of libglrtx.so (include/glrtx.h) -- the HIP device layer.

There is no CPU fallback: if the library is missing or no gfx950 device is
present, construction raises.  Tests and bench call the device path only
through this C ABI.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .host import LIB_DIR, PKG_ROOT

GLRTX_OK = 0
GLRTX_EINVAL, GLRTX_EDEVICE, GLRTX_ESCENE, GLRTX_EDEPTH, GLRTX_ENOMEM = -1, -2, -3, -4, -5
EXT_DIELECTRIC, EXT_WHITTED = 1, 2
FALLBACK_DEPTH, FALLBACK_SAMPLES, FALLBACK_EXTENSIONS = 1, 2, 4


class Params(C.Structure):
    _fields_ = [("c2w", C.c_float * 16), ("s2c", C.c_float * 16), ("aperture", C.c_float), ("focal", C.c_float),
                ("seed", C.c_float * 2), ("n_samples", C.c_int32), ("max_depth", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("rays_untraced", C.c_uint64), ("paths", C.c_uint64), ("launches", C.c_uint64), ("kernel_launches", C.c_uint64),
                ("kernel_ms_total", C.c_double), ("accumulate_ms_total", C.c_double), ("kernel_ms_last", C.c_float),
                ("frames_last", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("owned_rows", C.c_int32),
                ("stack_entries", C.c_int32), ("lds_bytes", C.c_int32), ("n_tri", C.c_int32), ("n_fork", C.c_int32),
                ("n_mat", C.c_int32), ("n_light", C.c_int32), ("variant_last", C.c_int32), ("fallback_last", C.c_int32),
                ("resolve_ms_last", C.c_float), ("node_fetch_last", C.c_int32), ("fallback_launches", C.c_uint64),
                ("pipe_slots", C.c_int32), ("pipe_resident_max", C.c_int32), ("device_error_pending", C.c_int32), ("wf_state_mib", C.c_int32),
                ("shadow_limited", C.c_int32), ("reserved2", C.c_int32), ("feed_launches", C.c_uint64), ("feed_appended", C.c_uint64)]


EXPORTS = ["glrtx_abi_version", "glrtx_create", "glrtx_destroy", "glrtx_last_error", "glrtx_upload_scene", "glrtx_build_lbvh", "glrtx_build_bvh_sah",
           "glrtx_resize", "glrtx_clear", "glrtx_set_partition", "glrtx_local_row_to_y", "glrtx_bind_accum",
           "glrtx_set_stream", "glrtx_set_variant", "glrtx_set_shadow_range_limit", "glrtx_count_rays", "glrtx_render", "glrtx_render_frames", "glrtx_sync", "glrtx_read_accum",
           "glrtx_accum_device_ptr", "glrtx_resolve_rgba8", "glrtx_get_stats", "glrtx_reset_stats",
           "glrtx_timer_begin", "glrtx_timer_end", "glrtx_upload_spheres", "glrtx_set_extensions",
           "glrtx_group_create", "glrtx_group_destroy", "glrtx_group_last_error", "glrtx_group_size", "glrtx_group_ctx",
           "glrtx_group_upload_scene", "glrtx_group_resize", "glrtx_group_clear", "glrtx_group_render", "glrtx_group_render_frames",
           "glrtx_debug_resolve_burst", "glrtx_hit_histogram", "glrtx_group_sync", "glrtx_group_read_accum", "glrtx_group_resolve_rgba8", "glrtx_group_get_stats", "glrtx_group_gather_copies"]

_lib = None


def lib_path():
    return LIB_DIR / "libglrtx.so"


def lib():
    global _lib
    if _lib is None:
        path = lib_path()
        if not path.exists():
            raise RuntimeError(f"{path} is missing: the HIP extension is not built "
                               f"(run `make -C {PKG_ROOT}` or __graft_entry__.build()); there is no fallback path")
        L = C.CDLL(str(path))
        vp, fp = C.c_void_p, C.POINTER(C.c_float)
        L.glrtx_abi_version.restype = C.c_int
        L.glrtx_create.argtypes = [C.POINTER(vp), C.c_int]
        L.glrtx_destroy.argtypes = [vp]
        L.glrtx_destroy.restype = None
        L.glrtx_last_error.argtypes = [vp]
        L.glrtx_last_error.restype = C.c_char_p
        L.glrtx_upload_scene.argtypes = [vp, fp, C.c_size_t, fp, C.c_size_t, fp, C.c_size_t, fp, C.c_size_t, fp,
                                         C.c_size_t]
        L.glrtx_build_lbvh.argtypes = [vp, fp, C.c_size_t, fp, C.c_size_t, fp, C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.glrtx_build_bvh_sah.argtypes = [vp, fp, C.c_size_t, fp, C.c_size_t, fp, C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.glrtx_resize.argtypes = [vp, C.c_int, C.c_int]
        L.glrtx_clear.argtypes = [vp]
        L.glrtx_set_partition.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.glrtx_local_row_to_y.argtypes = [vp, C.c_int]
        L.glrtx_bind_accum.argtypes = [vp, vp, C.c_size_t, C.c_int]
        L.glrtx_set_stream.argtypes = [vp, vp]
        L.glrtx_set_variant.argtypes = [vp, C.c_int]
        L.glrtx_set_shadow_range_limit.argtypes = [vp, C.c_int]
        L.glrtx_count_rays.argtypes = [vp, C.c_int]
        L.glrtx_render.argtypes = [vp, C.POINTER(Params)]
        L.glrtx_render_frames.argtypes = [vp, C.POINTER(Params), fp, C.c_int]
        L.glrtx_sync.argtypes = [vp]
        L.glrtx_read_accum.argtypes = [vp, vp, C.c_size_t]
        L.glrtx_accum_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.glrtx_resolve_rgba8.argtypes = [vp, vp, C.c_size_t, C.c_float, C.c_int]
        L.glrtx_get_stats.argtypes = [vp, C.POINTER(Stats)]
        try:  # (ABI 10; tools/gpu_abx.py also loads libraries of earlier rounds for A/B runs)
            L.glrtx_debug_resolve_burst.argtypes = [vp, C.c_float, C.c_int, C.POINTER(C.c_float)]
            L.glrtx_hit_histogram.argtypes = [vp, C.POINTER(Params), C.POINTER(C.c_uint32), C.c_size_t]
        except AttributeError:
            pass
        L.glrtx_reset_stats.argtypes = [vp]
        L.glrtx_timer_begin.argtypes = [vp]
        L.glrtx_timer_end.argtypes = [vp, C.POINTER(C.c_float)]
        L.glrtx_upload_spheres.argtypes = [vp, fp, C.c_size_t]
        L.glrtx_set_extensions.argtypes = [vp, C.c_int]
        L.glrtx_group_create.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]
        L.glrtx_group_destroy.argtypes = [vp]
        L.glrtx_group_destroy.restype = None
        L.glrtx_group_last_error.argtypes = [vp]
        L.glrtx_group_last_error.restype = C.c_char_p
        L.glrtx_group_size.argtypes = [vp]
        L.glrtx_group_ctx.argtypes = [vp, C.c_int]
        L.glrtx_group_ctx.restype = vp
        L.glrtx_group_upload_scene.argtypes = [vp, fp, C.c_size_t, fp, C.c_size_t, fp, C.c_size_t, fp, C.c_size_t, fp, C.c_size_t]
        L.glrtx_group_resize.argtypes = [vp, C.c_int, C.c_int]
        L.glrtx_group_clear.argtypes = [vp]
        L.glrtx_group_render.argtypes = [vp, C.POINTER(Params)]
        L.glrtx_group_render_frames.argtypes = [vp, C.POINTER(Params), fp, C.c_int]
        L.glrtx_group_sync.argtypes = [vp]
        L.glrtx_group_read_accum.argtypes = [vp, vp, C.c_size_t]
        L.glrtx_group_resolve_rgba8.argtypes = [vp, vp, C.c_size_t, C.c_float, C.c_int]
        L.glrtx_group_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.glrtx_group_gather_copies.argtypes = [vp]
        _lib = L
    return _lib


class GlrtxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"glrtx error {code}: {msg}")
        self.code = code


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def make_params(p) -> Params:
    r = Params()
    r.c2w[:] = list(_f32(p["c2w"]).reshape(16))
    r.s2c[:] = list(_f32(p["s2c"]).reshape(16))
    r.aperture = p.get("aperture", 0.0)
    r.focal = p.get("focal", 1.0)
    r.seed[:] = [p["seed"][0], p["seed"][1]]
    r.n_samples = int(p["n_samples"])
    r.max_depth = int(p["max_depth"])
    return r


class Device:
    """One glrtx_ctx: one GPU, one row-stripe partition of the image."""

    def __init__(self, device_id: int = -1):
        self.L = lib()
        self.h = C.c_void_p()
        rc = self.L.glrtx_create(C.byref(self.h), device_id)
        if rc != 0:
            raise GlrtxError(rc, self.L.glrtx_last_error(None).decode())

    def close(self):
        if self.h:
            self.L.glrtx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise GlrtxError(rc, self.L.glrtx_last_error(self.h).decode())

    def upload_scene(self, scene):
        v, t, m, l, b = (_f32(scene[k]) for k in ("vert", "tri", "mat", "light", "bvh"))
        self._ck(self.L.glrtx_upload_scene(self.h, _fp(v), v.size // 15, _fp(t), t.size // 4, _fp(m), m.size // 18,
                                           _fp(l), l.size // 4, _fp(b), b.size // 9))

    def build_lbvh(self, vert, tri):
        """Linear BVH built on the GPU.  Returns (nodes (n_nodes*3, 3) float32 in the wire format, max_depth, device ms)."""
        v, t = _f32(vert).reshape(-1, 15), _f32(tri).reshape(-1, 4)
        nodes = np.zeros(((2 * t.shape[0] - 1) * 3, 3), np.float32)
        depth, ms = C.c_int(0), C.c_float(0)
        self._ck(self.L.glrtx_build_lbvh(self.h, _fp(v), v.shape[0], _fp(t), t.shape[0], _fp(nodes), C.byref(depth), C.byref(ms)))
        return nodes, int(depth.value), float(ms.value)

    def build_bvh_sah(self, vert, tri):
        """Binned SAH by levels + exact sweep at the bottom, built on the GPU (glrtx_build_bvh_sah).  Returns (nodes, max_depth, device ms) like build_lbvh."""
        v, t = _f32(vert).reshape(-1, 15), _f32(tri).reshape(-1, 4)
        nodes = np.zeros(((2 * t.shape[0] - 1) * 3, 3), np.float32)
        depth, ms = C.c_int(0), C.c_float(0)
        self._ck(self.L.glrtx_build_bvh_sah(self.h, _fp(v), v.shape[0], _fp(t), t.shape[0], _fp(nodes), C.byref(depth), C.byref(ms)))
        return nodes, int(depth.value), float(ms.value)

    def upload_spheres(self, spheres):
        """EXTENSION (parity unpinned): (n, 5) rows [cx, cy, cz, radius, material]; None or empty removes them."""
        sp = _f32(np.zeros((0, 5)) if spheres is None else np.asarray(spheres, np.float32).reshape(-1, 5))
        self._ck(self.L.glrtx_upload_spheres(self.h, _fp(sp), sp.shape[0]))

    def set_extensions(self, flags: int):
        """EXTENSION (parity unpinned): EXT_DIELECTRIC | EXT_WHITTED."""
        self._ck(self.L.glrtx_set_extensions(self.h, int(flags)))

    def set_partition(self, rank, world, stripe_rows=16):
        self._ck(self.L.glrtx_set_partition(self.h, rank, world, stripe_rows))

    def resize(self, w, h):
        self._ck(self.L.glrtx_resize(self.h, w, h))

    def clear(self):
        self._ck(self.L.glrtx_clear(self.h))

    def bind_accum(self, device_ptr, pitch_bytes, capacity_rows):
        self._ck(self.L.glrtx_bind_accum(self.h, C.c_void_p(device_ptr), pitch_bytes, int(capacity_rows)))

    def set_stream(self, hip_stream):
        self._ck(self.L.glrtx_set_stream(self.h, C.c_void_p(hip_stream)))

    def set_variant(self, variant: int):
        self._ck(self.L.glrtx_set_variant(self.h, int(variant)))

    def set_shadow_range_limit(self, enable: bool):
        """False (default): shadow rays are searched like the reference's; True: with the range limit of rounds 1-4 (faster, not part of the bit-exact contract)."""
        self._ck(self.L.glrtx_set_shadow_range_limit(self.h, int(enable)))

    def count_rays(self, enable=True):
        self._ck(self.L.glrtx_count_rays(self.h, int(enable)))

    def render(self, params):
        p = params if isinstance(params, Params) else make_params(params)
        self._ck(self.L.glrtx_render(self.h, C.byref(p)))

    def render_frames(self, params, seeds):
        """Frames in flight: len(seeds) consecutive frames that differ only in u_seed, as one launch."""
        p = params if isinstance(params, Params) else make_params(dict(params, seed=(0.0, 0.0)) if "seed" not in params else params)
        sd = _f32(np.asarray(seeds, np.float32).reshape(-1, 2))
        self._ck(self.L.glrtx_render_frames(self.h, C.byref(p), _fp(sd), sd.shape[0]))

    def sync(self):
        self._ck(self.L.glrtx_sync(self.h))

    def stats(self) -> Stats:
        s = Stats()
        self._ck(self.L.glrtx_get_stats(self.h, C.byref(s)))
        return s

    def reset_stats(self):
        self._ck(self.L.glrtx_reset_stats(self.h))

    def local_rows_y(self):
        n = self.stats().owned_rows
        return np.array([self.L.glrtx_local_row_to_y(self.h, r) for r in range(n)], np.int64)

    def read_accum(self) -> np.ndarray:
        s = self.stats()
        out = np.zeros((s.owned_rows, s.width, 4), np.float32)
        self._ck(self.L.glrtx_read_accum(self.h, out.ctypes.data, s.width * 16))
        return out

    def resolve_rgba8(self, gamma=2.2, flip_y=True) -> np.ndarray:
        s = self.stats()
        out = np.zeros((s.owned_rows, s.width, 4), np.uint8)
        self._ck(self.L.glrtx_resolve_rgba8(self.h, out.ctypes.data, s.width * 4, gamma, int(flip_y)))
        return out

    def hit_histogram(self, params, n_tri) -> np.ndarray:
        """Closest hits per triangle of the uploaded scene in ONE calibration frame of `params` (glrtx_hit_histogram): input of host.order_by_hits."""
        p = params if isinstance(params, Params) else make_params(params)
        out = np.zeros(int(n_tri), np.uint32)
        self._ck(self.L.glrtx_hit_histogram(self.h, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_uint32)), int(n_tri)))
        return out

    def resolve_burst_ms(self, gamma=2.2, reps=32) -> float:
        """Device time of one launch of the resolve kernel, from `reps` launches back to back (glrtx_debug_resolve_burst)."""
        ms = C.c_float(0)
        self._ck(self.L.glrtx_debug_resolve_burst(self.h, gamma, int(reps), C.byref(ms)))
        return float(ms.value)

    def timer_begin(self):
        self._ck(self.L.glrtx_timer_begin(self.h))

    def timer_end(self) -> float:
        ms = C.c_float(0)
        self._ck(self.L.glrtx_timer_end(self.h, C.byref(ms)))
        return float(ms.value)


class Group:
    """glrtx_group: one context per listed HIP device (an ordinal may repeat), the image rows in interleaved 8-row stripes;
    read_accum / resolve_rgba8 return the FULL image, gathered on the first device."""

    def __init__(self, device_ids):
        self.L = lib()
        self.h = C.c_void_p()
        ids = (C.c_int * len(device_ids))(*device_ids)
        rc = self.L.glrtx_group_create(C.byref(self.h), ids, len(device_ids))
        if rc != 0:
            raise GlrtxError(rc, self.L.glrtx_group_last_error(None).decode())
        self.w = self.hgt = 0

    def close(self):
        if self.h:
            self.L.glrtx_group_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise GlrtxError(rc, self.L.glrtx_group_last_error(self.h).decode())

    def size(self):
        return int(self.L.glrtx_group_size(self.h))

    def member_call(self, fn, *args):
        """fn(ctx, *args) on every member (glrtx_count_rays, glrtx_set_variant, ...)."""
        for i in range(self.size()):
            rc = fn(self.L.glrtx_group_ctx(self.h, i), *args)
            if rc != 0:
                raise GlrtxError(rc, self.L.glrtx_last_error(self.L.glrtx_group_ctx(self.h, i)).decode())

    def upload_scene(self, scene):
        v, t, m, l, b = (_f32(scene[k]) for k in ("vert", "tri", "mat", "light", "bvh"))
        self._ck(self.L.glrtx_group_upload_scene(self.h, _fp(v), v.size // 15, _fp(t), t.size // 4, _fp(m), m.size // 18,
                                                 _fp(l), l.size // 4, _fp(b), b.size // 9))

    def resize(self, w, h):
        self._ck(self.L.glrtx_group_resize(self.h, w, h))
        self.w, self.hgt = w, h

    def clear(self):
        self._ck(self.L.glrtx_group_clear(self.h))

    def render(self, params):
        p = params if isinstance(params, Params) else make_params(params)
        self._ck(self.L.glrtx_group_render(self.h, C.byref(p)))

    def render_frames(self, params, seeds):
        p = params if isinstance(params, Params) else make_params(dict(params, seed=(0.0, 0.0)) if "seed" not in params else params)
        sd = _f32(np.asarray(seeds, np.float32).reshape(-1, 2))
        self._ck(self.L.glrtx_group_render_frames(self.h, C.byref(p), _fp(sd), sd.shape[0]))

    def sync(self):
        self._ck(self.L.glrtx_group_sync(self.h))

    def stats(self) -> Stats:
        s = Stats()
        self._ck(self.L.glrtx_group_get_stats(self.h, C.byref(s)))
        return s

    def read_accum(self) -> np.ndarray:
        out = np.zeros((self.hgt, self.w, 4), np.float32)
        self._ck(self.L.glrtx_group_read_accum(self.h, out.ctypes.data, self.w * 16))
        return out

    def gather_copies(self) -> int:
        return int(self.L.glrtx_group_gather_copies(self.h))

    def resolve_rgba8(self, gamma=2.2, flip_y=True) -> np.ndarray:
        out = np.zeros((self.hgt, self.w, 4), np.uint8)
        self._ck(self.L.glrtx_group_resolve_rgba8(self.h, out.ctypes.data, self.w * 4, gamma, int(flip_y)))
        return out


def render_image(scene, params, device_id=-1, count_rays=True):
    """Convenience: one pass from cleared accumulators on one GPU. Returns (accum (H,W,4), rays, kernel_ms)."""
    d = Device(device_id)
    try:
        d.upload_scene(scene)
        d.resize(params["width"], params["height"])
        d.count_rays(count_rays)
        d.render(params)
        d.sync()
        st = d.stats()
        return d.read_accum(), int(st.rays), float(st.kernel_ms_last)
    finally:
        d.close()
