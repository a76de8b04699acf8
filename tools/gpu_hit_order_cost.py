"""The measured child order by hits PER UNIT COST (glrt_bvh_order_by_hits with GLRT_HITS_COST_EXP=e: a subtree of n triangles is charged n^e; 0 = hits alone, round 6's first
form), per config, inside one context, trees alternating launch by launch.

    python tools/gpu_hit_order_cost.py [configs ...] [--exps 0,0.5,1,1.5] [--base sah|sah-reinsert]      (profiles/r06_hit_order_cost.txt)"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, "."); sys.path.insert(0, "opengl-raytracer_amd/python")
from glrt_amd import device, host, scenes  # noqa: E402

args = sys.argv[1:]
exps, base = ["0", "0.5", "1", "1.5"], "sah"
if "--exps" in args: i = args.index("--exps"); exps = args[i + 1].split(","); del args[i:i + 2]
if "--base" in args: i = args.index("--base"); base = args[i + 1]; del args[i:i + 2]
cfgs = args or ["headline", "c2", "c4", "c5"]
d = device.Device()
for name in cfgs:
    sc, pr = scenes.CONFIGS[name]() if name != "c4" else scenes.CONFIGS[name](n_samples=1)
    W, H = pr["width"], pr["height"]
    s = scenes.rebuild_bvh(sc, base)
    trees = {base: (s["bvh"], 0)}
    d.upload_scene(s); d.resize(480, 270)
    hist = d.hit_histogram(dict(pr, width=480, height=270, seed=host.frame_seed(12345)), s["tri"].shape[0])
    for e in exps:
        os.environ["GLRT_HITS_COST_EXP"] = e
        trees[f"{base}+hits, e={e}"] = host.order_by_hits(s["bvh"], hist, s["tri"], s["mat"])
    os.environ.pop("GLRT_HITS_COST_EXP", None)
    F, rounds = 16, 9
    ms = {k: [] for k in trees}
    sig = {}
    names = list(trees)
    for rnd in range(rounds + 1):
        for k in (names if rnd % 2 == 0 else names[::-1]):
            nodes = trees[k][0]
            d.upload_scene(dict(sc, bvh=nodes)); d.resize(W, H); d.clear()
            if rnd == 0:
                d.count_rays(True); d.reset_stats()
            d.render_frames(pr, [host.frame_seed(100 * rnd + i) for i in range(F)]); d.sync()
            if rnd == 0:
                sig[k] = (int(d.stats().rays), hashlib.sha1(np.ascontiguousarray(d.read_accum()).view(np.uint8)).hexdigest()[:12]); d.count_rays(False)
            else:
                ms[k].append(d.stats().kernel_ms_last / F)
    base_ms = float(np.median(ms[base]))
    print(f"== {name} {W}x{H}, {F} frames per launch, {rounds} rounds", flush=True)
    for k in names:
        m = float(np.median(ms[k]))
        print(f"  {k:28s} {m:.4f} ms/frame ({(m / base_ms - 1) * 100:+.2f} %)  forks exchanged {trees[k][1]:6d}  {'same image' if sig[k] == sig[base] else 'image differs (ties)'}", flush=True)
