#!/usr/bin/env python3
"""B1 of BASELINE.md: time the reference's UNMODIFIED shader on Mesa llvmpipe in the build container
(8 vCPU, LP_NUM_THREADS default) for the BASELINE configs, next to the C restatement (oracle) on the same cores.
Writes profiles/reference_llvmpipe_timing.json.  The reference cannot travel to the GPU box."""
import json, os, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "opengl-raytracer_amd" / "python"))
import numpy as np
from glrt_amd import scenes, host
from oracle import pt_oracle
from oracle.glref import GLRef
g = GLRef()
out = {"renderer": g.info(), "host_cpus": os.cpu_count(), "lp_num_threads": os.environ.get("LP_NUM_THREADS", "default (all cores)"), "configs": {}}
cases = [("c1", dict(), 8), ("headline", dict(), 2), ("c2", dict(), 2), ("c3", dict(width=480, height=270), 1), ("c5", dict(), 2)]
for name, kw, frames in cases:
    sc, pr = scenes.CONFIGS[name](**kw)
    g.render_reference(sc, dict(pr, seed=host.frame_seed(0)))  # warm-up: JIT of the fragment variant
    t = time.perf_counter()
    for f in range(frames):
        rgb, cnt = g.render_reference(sc, dict(pr, seed=host.frame_seed(f + 1)))
    gl_ms = (time.perf_counter() - t) / frames * 1e3
    acc = np.zeros((pr["height"], pr["width"], 4), np.float32)
    t = time.perf_counter(); rays = 0
    for f in range(frames):
        _, r = pt_oracle.render(sc, dict(pr, seed=host.frame_seed(f + 1)), accum=acc); rays += r
    or_ms = (time.perf_counter() - t) / frames * 1e3
    out["configs"][name] = {"size": [pr["width"], pr["height"]], "max_depth": pr["max_depth"], "triangles": int(sc["tri"].shape[0]),
                            "bvh": sc["bvh_kind"], "frames_timed": frames, "gl_reference_ms_per_frame": round(gl_ms, 1),
                            "oracle_port_ms_per_frame": round(or_ms, 1), "rays_per_frame": rays // frames,
                            "gl_reference_mrays_per_s": round(rays / frames / gl_ms / 1e3, 3)}
    print(name, out["configs"][name], flush=True)
(ROOT / "profiles" / "reference_llvmpipe_timing.json").write_text(json.dumps(out, indent=1))
