"""Soak of the shadow rays' search on hostile scenes (tests/fuzz_scenes.py: shadow_hostile): per seed the device's default (exact) search and the opt-in range
limit (glrtx_set_shadow_range_limit, csrc/pt_kernel.hip.h: shadow_limit) against the oracle.  The default must equal the oracle on every seed (exit status 1
otherwise); a seed on which the LIMITED search differs is a found instance of what makes the limit opt-in -- printed with its pixel count, not an error.

    python tools/gpu_shadow_fuzz.py FIRST COUNT [--size 96x64]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
from fuzz_scenes import shadow_hostile  # noqa: E402
from glrt_amd import device  # noqa: E402
from oracle import pt_oracle  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
w, h = 48, 32
if "--size" in sys.argv:
    w, h = (int(v) for v in sys.argv[sys.argv.index("--size") + 1].split("x"))
d = device.Device()
bad_default, bad_limit, rays = [], [], 0
for seed in range(first, first + count):
    tag, scene, params = shadow_hostile(seed, w, h)
    ref, n = pt_oracle.render(scene, params)
    rays += n
    d.upload_scene(scene); d.resize(w, h); d.count_rays(True)
    res = []
    for limited in (0, 1):
        d.set_shadow_range_limit(limited); d.clear(); d.reset_stats()
        d.render(params); d.sync()
        a = d.read_accum()
        same = (a.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(a) & np.isnan(ref))
        res.append(int((~same).any(-1).sum()))
    if res[0]:
        bad_default.append(seed); print(f"DEFAULT SEARCH DIFFERS: {tag}: {res[0]} pixels", flush=True)
    if res[1]:
        bad_limit.append(seed); print(f"range limit differs: {tag}: {res[1]} pixels", flush=True)
    if (seed - first) % 200 == 199:
        print(f"... {seed - first + 1} seeds, {rays} rays; default differs on {len(bad_default)}, range limit on {len(bad_limit)}", flush=True)
d.set_shadow_range_limit(0)
print(f"seeds {first} .. {first + count - 1} at {w}x{h}: {rays} rays; default (exact) search differs from the oracle on {len(bad_default)} seeds {bad_default[:20]}; "
      f"opt-in range limit differs on {len(bad_limit)} seeds {bad_limit[:20]}")
sys.exit(1 if bad_default else 0)
