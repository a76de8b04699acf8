"""What a launch costs at the cadences a host can drive it at, inside ONE context, settings alternating pass by pass.

    python tools/gpu_launch_cadence.py [--lib opengl-raytracer_amd/lib/libglrtx.so] [--config headline] [--passes 5] [NAME:ENV=VAL,ENV=VAL ...]

Per setting (default: one setting "default" with no overrides; an ENV=VAL pair is put into the process environment before each of the setting's passes -- the library
reads its GLRTX_* switches at every launch), per pass:
  lone20 / lone48   kernel time of ONE glrtx_render_frames launch of 20 / 48 frames, device idle before and after       (ms per frame)
  b2b16             wall time of 6 back-to-back glrtx_render_frames launches of 16 frames, one sync at the end           (ms per frame)
  one_per_frame     wall time of 96 back-to-back glrtx_render calls, one sync at the end (window.cpp:121-169's cadence)  (ms per frame)
  sync_per_frame    wall time of 32 glrtx_render calls with a glrtx_sync behind each (a host that looks at every frame)   (ms per frame)
and the image of the whole sequence (sha1 of the accumulator) -- every setting must give the same one.  Prints the medians over the passes."""
import hashlib
import os
import pathlib
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
from glrt_amd import device, host, scenes  # noqa: E402

a = sys.argv[1:]
opt = dict(lib="opengl-raytracer_amd/lib/libglrtx.so", config="headline", passes=5)
settings = []
while a:
    x = a.pop(0)
    if x.startswith("--"): opt[x[2:]] = type(opt[x[2:]])(a.pop(0))
    else:
        name, _, rest = x.partition(":")
        settings.append((name, dict(kv.split("=", 1) for kv in rest.split(",") if kv)))
settings = settings or [("default", {})]
device.lib_path = lambda: pathlib.Path(os.path.join(ROOT, opt["lib"]))
sc, pr = scenes.CONFIGS[opt["config"]]()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
all_env = sorted({k for _, e in settings for k in e})


def apply(env):
    for k in all_env:
        if k in env: os.environ[k] = env[k]
        else: os.environ.pop(k, None)


def seeds(f0, n): return [host.frame_seed(f0 + i) for i in range(n)]


def one_pass():
    out = {}
    f = 0
    d.clear()
    for n in (20, 48):
        d.sync(); d.render_frames(pr, seeds(f, n)); d.sync(); f += n
        out[f"lone{n}"] = d.stats().kernel_ms_last / n
    d.sync(); t0 = time.perf_counter()
    for k in range(6): d.render_frames(pr, seeds(f, 16)); f += 16
    d.sync(); out["b2b16"] = (time.perf_counter() - t0) * 1e3 / 96
    d.sync(); t0 = time.perf_counter()
    for k in range(96): d.render(dict(pr, seed=host.frame_seed(f))); f += 1
    d.sync(); out["one_per_frame"] = (time.perf_counter() - t0) * 1e3 / 96
    d.sync(); t0 = time.perf_counter()
    for k in range(32): d.render(dict(pr, seed=host.frame_seed(f))); d.sync(); f += 1
    out["sync_per_frame"] = (time.perf_counter() - t0) * 1e3 / 32
    out["sha1"] = hashlib.sha1(np.ascontiguousarray(d.read_accum()).view(np.uint8)).hexdigest()[:16]
    return out


res = {n: [] for n, _ in settings}
apply(settings[0][1]); one_pass()  # warm-up: buffers allocated, code objects loaded
for p in range(opt["passes"]):
    order = settings if p % 2 == 0 else settings[::-1]
    for name, env in order:
        apply(env)
        res[name].append(one_pass())
first = None
for name, env in settings:
    r = res[name]
    first = first or r[0]["sha1"]
    med = {k: float(np.median([x[k] for x in r])) for k in ("lone20", "lone48", "b2b16", "one_per_frame", "sync_per_frame")}
    same = "same image" if all(x["sha1"] == first for x in r) else "IMAGE DIFFERS"
    print(f"{name:24s} lone20 {med['lone20']:.4f}  lone48 {med['lone48']:.4f}  b2b16 {med['b2b16']:.4f}  one_per_frame {med['one_per_frame']:.4f}  sync_per_frame {med['sync_per_frame']:.4f} ms/frame   {same}   {env}", flush=True)
st = d.stats()
print(f"pipe_slots {st.pipe_slots} pipe_resident_max {st.pipe_resident_max} wf_state_mib {st.wf_state_mib}", flush=True)
