"""Diagnostic for one scale of tools/gpu_scale_sweep.py: which pixels differ from the oracle, per kernel variant and fetch form.
   python tools/gpu_scale_probe.py LIB SCALE [depth]"""
import os, sys, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
import numpy as np
from glrt_amd import device, scenes
from oracle import pt_oracle
device.lib_path = lambda: pathlib.Path(os.path.join(ROOT, sys.argv[1]))
k = float(sys.argv[2]); depth = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sc0, pr0 = scenes.config_c1(48, 32, max_depth=depth, n_samples=1, bvh="sah", subdiv=1)
kf = np.float32(k)
vert = sc0["vert"].reshape(-1, 5, 3).copy(); vert[:, 0] *= kf
nodes = sc0["bvh"].reshape(-1, 9).copy(); nodes[:, 0:6] *= kf
sc = dict(sc0, vert=vert.reshape(-1, 3), bvh=nodes.reshape(-1, 3))
c2w = np.array(pr0["c2w"], np.float32).reshape(4, 4).copy(); c2w[3, :3] *= kf
p = dict(pr0, c2w=c2w.reshape(-1))
ref, rays = pt_oracle.render(sc, p)
d = device.Device(); d.upload_scene(sc); d.resize(48, 32)
for variant in (2, 1, 0):
    for fetch in ("0", "1", "2"):
        if variant != 2 and fetch != "0": continue
        os.environ["GLRTX_PAIR_FETCH"] = fetch
        d.set_variant(variant); d.clear(); d.count_rays(True); d.reset_stats(); d.render(p); d.sync()
        acc = d.read_accum()
        diff = (acc.view(np.uint32) != ref.view(np.uint32)).any(-1)
        ys, xs = np.nonzero(diff)
        print(f"variant {variant} fetch {fetch}: {int(diff.sum())} pixels differ, rays {d.stats().rays} / {rays}", [(int(y), int(x), acc[y, x].tolist(), ref[y, x].tolist()) for y, x in zip(ys[:3], xs[:3])], flush=True)
