"""This repository's own GLSL statement of the path tracer (oracle/glsl/pt_port.frag) on Mesa llvmpipe, through oracle/glref.

TEST INFRASTRUCTURE / CPU BASELINE ONLY -- never imported by the product path.

Why it exists: BASELINE.json wants the reference "timed through Mesa llvmpipe on the box's own host cores in the same run".
The reference's shader file cannot travel to the GPU box; Mesa's swrast_dri.so and oracle/_ref/libglref.so are there.  So the
llvmpipe leg of bench.py's cpu_baseline runs THIS shader -- written from oracle/pt_oracle.c, and checked in the build container
(tests/test_glsl_port.py) on llvmpipe against the golden images the reference's unmodified shader produced.

    python -m oracle.glport --config headline --frames 3 --warmup 1        (prints one JSON line; run as a CHILD process by bench.py:
                                                                            llvmpipe's LLVM stays out of the process that holds the GPU)
"""
from __future__ import annotations

import ctypes as C
import json
import os
import pathlib
import sys
import time

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
PORT_FS = _HERE / "glsl" / "pt_port.frag"


def available() -> bool:
    from oracle import glref
    return glref.available() and PORT_FS.exists()


def render(scene, params, frames=None, timings=None):
    """Own GLSL port on llvmpipe: one pass per seed in `frames` (default: params['seed'] once), ping-pong accumulated from cleared
    targets (window.cpp:214-252; the previous frame is fetched exactly, not through a LINEAR sampler -- SURVEY.md F7).
    Returns (rgb (h, w, 3), count (h, w)) float32, row 0 = bottom row.  timings: a list that receives seconds per pass (draw + glFinish)."""
    from oracle import glref
    g = glref.GLRef()
    L = g.L
    p = g.program(glref.DIAG_VS, PORT_FS.read_text())
    L.glref_use(p)
    w, h = int(params["width"]), int(params["height"])
    bufs = [("vertTex", scene["vert"], 3, 2), ("triTex", scene["tri"], 4, 3), ("matTex", scene["mat"], 3, 4),
            ("lightTex", scene["light"], 4, 5), ("nodeTex", scene["bvh"], 3, 6)]
    handles = []
    for name, arr, comps, unit in bufs:
        a = np.asarray(arr, np.float32)
        if a.size == 0:  # (a scene without lights: an empty texture buffer cannot be created; one zero texel is never fetched with lightCount 0 ... -1 clamps to it)
            a = np.zeros((1, comps), np.float32)
        hdl = g.tbo(a, comps)
        handles.append(hdl)
        L.glref_bind_tbo(unit, hdl[0])
        L.glref_uniform1i(p, name.encode(), unit)
    n_lights = int(np.asarray(scene["light"]).reshape(-1, 4).shape[0])
    g._set_uniforms(p, {
        "camToWorld": params["c2w"], "screenToCam": params["s2c"],
        "lensRadius": float(params.get("aperture", 0.0)), "focusDist": float(params.get("focal", 1.0)),
        "samplesPerPass": int(params["n_samples"]), "depthLimit": int(params["max_depth"]),
        "imageSize": (float(w), float(h)), "lightCount": n_lights,
    })
    z3, z1 = np.zeros((h, w, 3), np.float32), np.zeros((h, w), np.float32)
    tex = [[L.glref_tex2d(w, h, 3, z3.ctypes.data), L.glref_tex2d(w, h, 1, z1.ctypes.data)] for _ in range(2)]
    fbo = [L.glref_fbo(2, (C.c_uint * 2)(*tex[i])) for i in range(2)]
    if not all(fbo):
        raise RuntimeError(g.err())
    sel = 0
    for sd in ([params["seed"]] if frames is None else list(frames)):
        sel ^= 1
        L.glref_uniform2f(p, b"frameSeed", float(sd[0]), float(sd[1]))
        L.glref_bind_tex2d(0, tex[sel ^ 1][0])
        L.glref_uniform1i(p, b"prevSum", 0)
        L.glref_bind_tex2d(1, tex[sel ^ 1][1])
        L.glref_uniform1i(p, b"prevCount", 1)
        t = time.perf_counter()
        if L.glref_draw(fbo[sel], w, h, 1) != 0:
            raise RuntimeError(g.err())
        if timings is not None:
            timings.append(time.perf_counter() - t)
    rgb = np.empty((h, w, 3), np.float32)
    cnt = np.empty((h, w), np.float32)
    L.glref_read_tex2d(tex[sel][0], 3, rgb.ctypes.data)
    L.glref_read_tex2d(tex[sel][1], 1, cnt.ctypes.data)
    for f in fbo:
        L.glref_fbo_free(f)
    for pair in tex:
        for t_ in pair:
            L.glref_tex_free(t_)
    for hdl in handles:
        L.glref_tbo_free(hdl[0], hdl[1])
    return rgb, cnt


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="headline")
    ap.add_argument("--frames", type=int, default=3, help="timed passes (frames of the accumulation loop)")
    ap.add_argument("--warmup", type=int, default=1, help="untimed passes in front (the first draw JIT-compiles the shader)")
    ap.add_argument("--first-frame", type=int, default=0, help="index of the first timed frame (host.frame_seed)")
    ap.add_argument("--threads", type=int, default=0, help="LP_NUM_THREADS (0: llvmpipe's default = all cores it sees, at most 16)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--check", action="store_true", help="compare the accumulated image with the C restatement (oracle/pt_oracle.c) run over the same frames")
    args = ap.parse_args(argv)
    if args.threads > 0:
        os.environ["LP_NUM_THREADS"] = str(args.threads)  # read by llvmpipe when the screen is created
    sys.path.insert(0, str(_HERE.parent))
    sys.path.insert(0, str(_HERE.parent / "opengl-raytracer_amd" / "python"))
    from glrt_amd import host, scenes
    from oracle import glref
    if not available():
        print(json.dumps({"available": False, "reason": "Mesa swrast_dri.so or oracle/_ref/libglref.so absent"}))
        return 0
    kw = {}
    if args.width and args.height:
        kw = dict(width=args.width, height=args.height)
    scene, params = scenes.CONFIGS[args.config](**kw)
    seeds = [host.frame_seed(args.first_frame + i) for i in range(args.frames)]
    if args.warmup > 0:
        render(scene, params, frames=[host.frame_seed(10_000 + i) for i in range(args.warmup)])
    times = []
    rgb, cnt = render(scene, params, frames=seeds, timings=times)
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    try:
        quota, period = pathlib.Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            cores = min(cores, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    lp = int(os.environ.get("LP_NUM_THREADS", "0")) or min(cores, 16)  # llvmpipe caps its rasteriser threads at 16 (LP_MAX_THREADS)
    out = {"available": True, "renderer": glref.GLRef().info(), "config": args.config, "width": params["width"], "height": params["height"],
           "frames": args.frames, "ms_per_frame": round(float(np.mean(times)) * 1e3, 2), "ms_per_frame_min": round(float(np.min(times)) * 1e3, 2),
           "cores": cores, "rasteriser_threads": lp, "shader": "oracle/glsl/pt_port.frag (this repository's own GLSL statement; the reference's file does not travel)"}
    if args.check:
        from oracle import pt_oracle
        acc, rays = None, 0
        for sd in seeds:
            acc, n = pt_oracle.render(scene, dict(params, seed=sd), accum=acc)
            rays += n
        a = np.concatenate([rgb, cnt[..., None]], -1).astype(np.float32)
        diff = int((a.view(np.uint32) != acc.view(np.uint32)).any(-1).sum())
        out["rays"] = rays
        out["image_vs_c_restatement"] = "bit-identical" if diff == 0 else f"{diff} of {a.shape[0] * a.shape[1]} pixels differ"
    print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
