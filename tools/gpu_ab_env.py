"""A/B of a run-time switch (an environment variable the library reads at every launch) inside ONE context: the same buffers, the same box and clock, the launches of the two
settings alternating.  This removes the context-to-context spread (profiles/r04_context_regimes.txt) from the comparison altogether: differences of 0.1-0.2 % show.

    python tools/gpu_ab_env.py LIB ENV VALUE_A VALUE_B [--contexts 3] [--rounds 30] [--frames 20] [--config headline|c2..c5|rand:N]
    python tools/gpu_ab_env.py LIB call:set_shadow_range_limit 0 1        (a per-context setter of glrt_amd.device.Device instead of an environment variable)

Per context: median ms per frame of each setting and the median (quartiles) of the per-round differences B against A; images and ray counts of the two settings compared first."""
import hashlib
import os
import pathlib
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
from glrt_amd import device, host, scenes  # noqa: E402

a = sys.argv[1:]
lib, env, va, vb = a[0], a[1], a[2], a[3]
opt = dict(contexts=3, rounds=30, frames=20, config="headline")
rest = a[4:]
while rest:
    k = rest.pop(0).lstrip("-"); opt[k] = type(opt[k])(rest.pop(0))
device.lib_path = lambda: pathlib.Path(os.path.join(ROOT, lib))


def setting(d, v):
    if env.startswith("call:"):
        getattr(d, env[5:])(int(v))
    else:
        os.environ[env] = v


if opt["config"].startswith("rand:"):  # rand:N -- N random triangles at config 5's density and camera distance (tree-size sweeps, as tools/gpu_abx.py)
    n_ = int(opt["config"][5:]); ext_ = 10.0 * (n_ / 100_000.0) ** (1.0 / 3.0)
    sc, pr = scenes._random_tri_scene(n_, 20260102, ext_, 3.4 * ext_, 1920, 1080, 4, 1, "sah")
else:
    sc, pr = scenes.CONFIGS[opt["config"]]()
F = opt["frames"]
for ci in range(opt["contexts"]):
    d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
    sig = []
    for v in (va, vb):
        setting(d, v)
        d.clear(); d.count_rays(True); d.reset_stats()
        d.render_frames(pr, [host.frame_seed(i) for i in range(F)]); d.sync()
        sig.append((int(d.stats().rays), hashlib.sha1(np.ascontiguousarray(d.read_accum()).view(np.uint8)).hexdigest()))
    d.count_rays(False)
    ms = {va: [], vb: []}
    r = 1
    for rnd in range(opt["rounds"] + 2):
        for v in ((va, vb) if rnd % 2 == 0 else (vb, va)):
            setting(d, v)
            d.render_frames(pr, [host.frame_seed(F * r + i) for i in range(F)]); d.sync(); r += 1
            if rnd >= 2: ms[v].append(d.stats().kernel_ms_last / F)
    A, B = np.asarray(ms[va]), np.asarray(ms[vb])
    q = np.percentile((B - A) / A * 100.0, [25, 50, 75])
    print(f"context {ci}: {env}={va} {np.median(A):.4f}  {env}={vb} {np.median(B):.4f} ms/frame; B against A {q[1]:+.2f} % (quartiles {q[0]:+.2f} .. {q[2]:+.2f}); images and rays {'identical' if sig[0] == sig[1] else 'DIFFER'}", flush=True)
    d.close()
