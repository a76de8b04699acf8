"""ctypes front-end for oracle/_ref/libglref.so (headless Mesa llvmpipe GL runner).

TEST INFRASTRUCTURE -- only usable in the build container (needs Mesa's
swrast_dri.so and, for `render_reference`, the reference checkout at
/root/reference).  Nothing under the product path imports this module.

`render_reference` executes the reference's *unmodified* fragment shader
(src/shaders/raytrace.frag, read as text at run time, never stored here) the
way the reference host drives it (window.cpp:213-295): same uniform names,
same texture units 0..6, one glDrawArrays(GL_TRIANGLES, 0, 6) into an
RGB32F + R32F framebuffer (window.cpp:366-381).  The accumulators start
cleared, so `u_nSamples = spp` gives a single-pass image.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_LIB = _HERE / "_ref" / "libglref.so"
REFERENCE_SHADERS = pathlib.Path(os.environ.get("GLRT_REFERENCE", "/root/reference")) / "src" / "shaders"

# Fullscreen pass for diagnostic programs (own code; two triangles from gl_VertexID).
DIAG_VS = """#version 410
void main() {
    vec2 p = vec2((gl_VertexID == 2 || gl_VertexID == 4 || gl_VertexID == 5) ? 1.0 : -1.0,
                  (gl_VertexID == 1 || gl_VertexID == 2 || gl_VertexID == 4) ? 1.0 : -1.0);
    gl_Position = vec4(p, 0.0, 1.0);
}
"""


def available() -> bool:
    return _LIB.exists() and os.path.exists("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so")


def reference_available() -> bool:
    return available() and (REFERENCE_SHADERS / "raytrace.frag").exists()


class GLRef:
    _inst = None

    def __new__(cls):
        if cls._inst is None:
            cls._inst = super().__new__(cls)
            cls._inst._init()
        return cls._inst

    def _init(self):
        L = C.CDLL(str(_LIB))
        L.glref_last_error.restype = C.c_char_p
        L.glref_string.restype = C.c_char_p
        L.glref_string.argtypes = [C.c_int]
        L.glref_program.restype = C.c_uint
        L.glref_program.argtypes = [C.c_char_p, C.c_char_p]
        L.glref_use.argtypes = [C.c_uint]
        L.glref_uniform1i.argtypes = [C.c_uint, C.c_char_p, C.c_int]
        L.glref_uniform1f.argtypes = [C.c_uint, C.c_char_p, C.c_float]
        L.glref_uniform2f.argtypes = [C.c_uint, C.c_char_p, C.c_float, C.c_float]
        L.glref_uniform3f.argtypes = [C.c_uint, C.c_char_p, C.c_float, C.c_float, C.c_float]
        L.glref_uniform_mat4.argtypes = [C.c_uint, C.c_char_p, C.c_void_p]
        L.glref_tbo.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint)]
        L.glref_tbo_free.argtypes = [C.c_uint, C.c_uint]
        L.glref_bind_tbo.argtypes = [C.c_int, C.c_uint]
        L.glref_tex2d.restype = C.c_uint
        L.glref_tex2d.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.glref_tex_free.argtypes = [C.c_uint]
        L.glref_bind_tex2d.argtypes = [C.c_int, C.c_uint]
        L.glref_read_tex2d.argtypes = [C.c_uint, C.c_int, C.c_void_p]
        L.glref_fbo.restype = C.c_uint
        L.glref_fbo.argtypes = [C.c_int, C.POINTER(C.c_uint)]
        L.glref_fbo_free.argtypes = [C.c_uint]
        L.glref_draw.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int]
        L.glref_tex2d_rgba8.restype = C.c_uint
        L.glref_tex2d_rgba8.argtypes = [C.c_int, C.c_int]
        L.glref_read_pixels_rgba8.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_void_p]
        self.L = L
        if L.glref_init() != 0:
            raise RuntimeError("glref_init: " + L.glref_last_error().decode())
        self.renderer = L.glref_string(0).decode()
        self.version = L.glref_string(1).decode()
        self._progs = {}

    # ------------------------------------------------------------------ utils
    def err(self) -> str:
        return self.L.glref_last_error().decode()

    def info(self) -> str:
        return f"{self.renderer}; GL {self.version}"

    def program(self, vs: str, fs: str) -> int:
        key = hash((vs, fs))
        if key not in self._progs:
            p = self.L.glref_program(vs.encode(), fs.encode())
            if not p:
                raise RuntimeError(self.err())
            self._progs[key] = p
        return self._progs[key]

    def tbo(self, arr: np.ndarray, comps: int):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        out = (C.c_uint * 2)()
        if self.L.glref_tbo(a.ctypes.data, a.nbytes, comps, out) != 0:
            raise RuntimeError("glref_tbo failed")
        return (out[0], out[1], a)

    # --------------------------------------------------------- generic passes
    def run_fragment(self, fs: str, w: int, h: int, out_comps=(4,), tbos=(), uniforms=None,
                     vs: str = DIAG_VS):
        """Run one fullscreen pass of fragment shader `fs`.

        tbos: sequence of (sampler_name, ndarray, comps) bound to units 0..;
        uniforms: {name: int | float | (2/3 floats) | 16-float matrix (column-major)}.
        Returns a list of float32 arrays (h, w, comps) for each colour attachment.
        """
        L = self.L
        p = self.program(vs, fs)
        L.glref_use(p)
        handles = []
        for unit, (name, arr, comps) in enumerate(tbos):
            hdl = self.tbo(arr, comps)
            handles.append(hdl)
            L.glref_bind_tbo(unit, hdl[0])
            L.glref_uniform1i(p, name.encode(), unit)
        self._set_uniforms(p, uniforms or {})
        texs = [L.glref_tex2d(w, h, c, None) for c in out_comps]
        arr_t = (C.c_uint * len(texs))(*texs)
        fbo = L.glref_fbo(len(texs), arr_t)
        if not fbo:
            raise RuntimeError(self.err())
        if L.glref_draw(fbo, w, h, 1) != 0:
            raise RuntimeError(self.err())
        outs = []
        for t, c in zip(texs, out_comps):
            o = np.empty((h, w, c), dtype=np.float32)
            L.glref_read_tex2d(t, c, o.ctypes.data)
            outs.append(o)
        L.glref_fbo_free(fbo)
        for t in texs:
            L.glref_tex_free(t)
        for hdl in handles:
            L.glref_tbo_free(hdl[0], hdl[1])
        return outs

    def _set_uniforms(self, p, uniforms):
        L = self.L
        for name, v in uniforms.items():
            n = name.encode()
            if isinstance(v, (bool, int, np.integer)):
                L.glref_uniform1i(p, n, int(v))
            elif isinstance(v, (float, np.floating)):
                L.glref_uniform1f(p, n, float(v))
            else:
                a = np.ascontiguousarray(v, dtype=np.float32).reshape(-1)
                if a.size == 2:
                    L.glref_uniform2f(p, n, a[0], a[1])
                elif a.size == 3:
                    L.glref_uniform3f(p, n, a[0], a[1], a[2])
                elif a.size == 16:
                    L.glref_uniform_mat4(p, n, a.ctypes.data)
                else:
                    raise ValueError(f"uniform {name}: unsupported size {a.size}")

    # --------------------------------------------- the reference's own shader
    def render_reference(self, scene, params, frames=None):
        """Run the reference's raytrace.{vert,frag} verbatim on `scene`.

        scene: dict with float32 arrays vert (nV*5,3) tri (nT,4) mat (nM*6,3) light (nL,4) bvh (nN*3,3).
        params: dict c2w(16, column-major) s2c(16) aperture focal seed(2) n_samples max_depth width height.
        frames: optional list of seeds; if given, ping-pong accumulate one pass per seed the way
                window.cpp:213-252 does (prev frame sampled through GL_LINEAR), else one pass from
                cleared accumulators.
        Returns (rgb (h,w,3), count (h,w)) float32, row 0 = bottom row (GL origin).
        """
        L = self.L
        vs = (REFERENCE_SHADERS / "raytrace.vert").read_text()
        fs = (REFERENCE_SHADERS / "raytrace.frag").read_text()
        p = self.program(vs, fs)
        L.glref_use(p)
        w, h = int(params["width"]), int(params["height"])
        bufs = [("u_vertBuffer", scene["vert"], 3, 2), ("u_triBuffer", scene["tri"], 4, 3),
                ("u_matBuffer", scene["mat"], 3, 4), ("u_lightBuffer", scene["light"], 4, 5),
                ("u_bvhBuffer", scene["bvh"], 3, 6)]
        handles = []
        for name, arr, comps, unit in bufs:
            hdl = self.tbo(arr, comps)
            handles.append(hdl)
            L.glref_bind_tbo(unit, hdl[0])
            L.glref_uniform1i(p, name.encode(), unit)
        n_lights = int(np.asarray(scene["light"]).reshape(-1, 4).shape[0])
        n_tris = int(np.asarray(scene["tri"]).reshape(-1, 4).shape[0])
        self._set_uniforms(p, {
            "u_c2wMat": params["c2w"], "u_s2cMat": params["s2c"],
            "u_apertureRadius": float(params.get("aperture", 0.0)),
            "u_focalLength": float(params.get("focal", 1.0)),
            "u_nSamples": int(params["n_samples"]), "u_maxDepth": int(params["max_depth"]),
            "u_windowSize": (float(w), float(h)),
            "u_nTris": n_tris, "u_nLights": n_lights, "u_hasVolume": 0,
        })
        # two ping-pong targets (window.cpp:366-381)
        z3, z1 = np.zeros((h, w, 3), np.float32), np.zeros((h, w), np.float32)
        tex = [[L.glref_tex2d(w, h, 3, z3.ctypes.data), L.glref_tex2d(w, h, 1, z1.ctypes.data)] for _ in range(2)]
        fbo = [L.glref_fbo(2, (C.c_uint * 2)(*tex[i])) for i in range(2)]
        if not all(fbo):
            raise RuntimeError(self.err())
        seeds = [params["seed"]] if frames is None else list(frames)
        sel = 0
        for sd in seeds:
            sel ^= 1
            L.glref_uniform2f(p, b"u_seed", float(sd[0]), float(sd[1]))
            L.glref_bind_tex2d(0, tex[sel ^ 1][0])
            L.glref_uniform1i(p, b"u_framebuffer", 0)
            L.glref_bind_tex2d(1, tex[sel ^ 1][1])
            L.glref_uniform1i(p, b"u_counter", 1)
            if L.glref_draw(fbo[sel], w, h, 1) != 0:
                raise RuntimeError(self.err())
        rgb = np.empty((h, w, 3), np.float32)
        cnt = np.empty((h, w), np.float32)
        L.glref_read_tex2d(tex[sel][0], 3, rgb.ctypes.data)
        L.glref_read_tex2d(tex[sel][1], 1, cnt.ctypes.data)
        for f in fbo:
            L.glref_fbo_free(f)
        for pair in tex:
            for t in pair:
                L.glref_tex_free(t)
        for hdl in handles:
            L.glref_tbo_free(hdl[0], hdl[1])
        return rgb, cnt

    # ------------------------------------------------- the reference's resolve pass
    def render_screen(self, rgb, count, gamma=None):
        """Run the reference's screen.{vert,frag} verbatim (src/shaders/screen.frag:15-25) the way Window::render drives it
        (window.cpp:297-317): u_framebuffer / u_counter = the accumulation targets (RGB32F + R32F, GL_LINEAR samplers as
        framebuffer_object.cpp creates them) on units 0 / 1, u_windowSize, one glDrawArrays(GL_TRIANGLES, 0, 6) into an RGBA8
        colour buffer, then saveCurrentFrame's glReadPixels(GL_RGBA, GL_UNSIGNED_BYTE) (window.cpp:383-388).
        gamma None leaves u_gamma at the shader's default (2.2), as the reference host does.
        Returns uint8 (h, w, 4), row 0 = bottom row (GL origin; the reference flips it before writing the PNG)."""
        L = self.L
        vs = (REFERENCE_SHADERS / "screen.vert").read_text()
        fs = (REFERENCE_SHADERS / "screen.frag").read_text()
        p = self.program(vs, fs)
        L.glref_use(p)
        rgb = np.ascontiguousarray(rgb, np.float32)
        count = np.ascontiguousarray(count, np.float32)
        h, w = count.shape
        t_rgb = L.glref_tex2d(w, h, 3, rgb.ctypes.data)
        t_cnt = L.glref_tex2d(w, h, 1, count.ctypes.data)
        L.glref_bind_tex2d(0, t_rgb)
        L.glref_uniform1i(p, b"u_framebuffer", 0)
        L.glref_bind_tex2d(1, t_cnt)
        L.glref_uniform1i(p, b"u_counter", 1)
        L.glref_uniform2f(p, b"u_windowSize", float(w), float(h))
        # (program objects are cached here and keep their uniform values: always set it; 2.2 is the shader's initialiser)
        L.glref_uniform1f(p, b"u_gamma", 2.2 if gamma is None else float(gamma))
        target = L.glref_tex2d_rgba8(w, h)
        fbo = L.glref_fbo(1, (C.c_uint * 1)(target))
        if not fbo:
            raise RuntimeError(self.err())
        if L.glref_draw(fbo, w, h, 1) != 0:
            raise RuntimeError(self.err())
        out = np.empty((h, w, 4), np.uint8)
        if L.glref_read_pixels_rgba8(fbo, w, h, out.ctypes.data) != 0:
            raise RuntimeError(self.err())
        L.glref_fbo_free(fbo)
        for t_ in (t_rgb, t_cnt, target):
            L.glref_tex_free(t_)
        return out
