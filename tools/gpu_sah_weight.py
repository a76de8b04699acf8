"""The CPU SAH builder's count weight (host/bvh.cpp: weight(); GLRT_SAH_ALPHA, 1 = the plain surface-area heuristic, default 0.8): one scene under the trees of several
exponents, rendered alternately inside one context (medians over the rounds), images compared.  Each tree is built in a child process (the exponent is read once per process).

    python tools/gpu_sah_weight.py [config] [frames per launch] [rounds] [exponents, comma-separated: the first is the baseline]      (profiles/r06_sah_count_weight.txt)"""
import os, subprocess, sys, pathlib, pickle
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "opengl-raytracer_amd" / "python"))
if len(sys.argv) > 1 and sys.argv[1] == "--build":
    from glrt_amd import scenes
    sc, pr = scenes.CONFIGS[sys.argv[2]]()
    pickle.dump((np.asarray(sc["bvh"]), int(sc["bvh_depth"])), open(sys.argv[3], "wb"))
    sys.exit(0)
from glrt_amd import device, host, scenes
cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
alphas = (sys.argv[4] if len(sys.argv) > 4 else "1,0.9,0.8,0.7,0.6,0.5,0.35").split(",")
sc, pr = scenes.CONFIGS[cfg]()
trees = {}
for a in alphas:
    f = f"/tmp/tree_{cfg}_{a.replace('/', '_')}.pkl"
    parts = a.split("@")  # "0.8" or "0.8@GLRT_SAH_BINS_TOP=32@..." (further builder switches for this variant)
    env = dict(os.environ, GLRT_SAH_ALPHA=parts[0], **dict(kv.split("=", 1) for kv in parts[1:]))
    subprocess.run([sys.executable, __file__, "--build", cfg, f], env=env, check=True)
    trees[a] = pickle.load(open(f, "rb"))
d = device.Device()
ms = {k: [] for k in trees}
names = list(trees)
import hashlib
sig = {}
for rnd in range(rounds + 1):
    for k in (names if rnd % 2 == 0 else names[::-1]):
        nodes, depth = trees[k]
        d.upload_scene(dict(sc, bvh=nodes, bvh_depth=depth)); d.resize(pr["width"], pr["height"])
        if rnd == 0:
            d.count_rays(True); d.reset_stats(); d.clear()
            d.render_frames(pr, [host.frame_seed(i) for i in range(2)]); d.sync()
            sig[k] = (int(d.stats().rays), hashlib.sha1(np.ascontiguousarray(d.read_accum()).view(np.uint8)).hexdigest()[:10], int(d.stats().stack_entries))
            d.count_rays(False)
            continue
        d.render_frames(pr, [host.frame_seed(100 * rnd + i) for i in range(F)]); d.sync()
        d.render_frames(pr, [host.frame_seed(100 * rnd + 50 + i) for i in range(F)]); d.sync()
        ms[k].append(d.stats().kernel_ms_last / F)
base = float(np.median(ms[names[0]]))
print(f"{cfg}: {F} frames per launch, {rounds} rounds, trees alternated inside one context")
for k in names:
    m = float(np.median(ms[k]))
    print(f"  alpha {k:34s} {m:8.4f} ms/frame ({(m / base - 1) * 100:+6.2f} %)  depth {trees[k][1]:3d}  stack {sig[k][2]:3d}  rays {sig[k][0]}  image {sig[k][1]}", flush=True)
