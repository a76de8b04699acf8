"""Experiment (round 5): insertion-based optimisation of the finished tree (glrt_bvh_reinsert, host/bvh.cpp) against the builder's tree, lights first on both, rendered
alternately in one context.
    python tools/gpu_reinsert.py [config] [frames per launch] [rounds] [max passes]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
import numpy as np
from glrt_amd import device, host, scenes

cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 10
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 8
sc, pr = scenes.CONFIGS[cfg]()
raw = sc["bvh_builder"]
opt, depth, moved, cost = host.reinsert(raw, passes)
def canonical(nodes):
    """children ordered as a top-down builder leaves them: the child whose box centre is lower on the axis where the two centres differ most goes into x"""
    N = np.array(nodes, np.float32).reshape(-1, 9).copy()
    for i in range(N.shape[0]):
        if N[i, 8] < 0:
            x, y = int(N[i, 6]), int(N[i, 7])
            cx, cy = (N[x, :3] + N[x, 3:6]) * 0.5, (N[y, :3] + N[y, 3:6]) * 0.5
            k = int(np.argmax(np.abs(cx - cy)))
            if cx[k] > cy[k]:
                N[i, 6], N[i, 7] = y, x
    return N.reshape(-1, 3)


trees = {"the builder's tree": host.lights_first(raw, sc["tri"], sc["mat"])[0],
         "the builder's tree, children in canonical order": host.lights_first(canonical(raw), sc["tri"], sc["mat"])[0],
         "reinserted, children in canonical order": host.lights_first(canonical(opt), sc["tri"], sc["mat"])[0],
         f"reinserted ({moved} subtrees moved, fork area {cost[0]:.2f} -> {cost[1]:.2f}, depth {depth})": host.lights_first(opt, sc["tri"], sc["mat"])[0]}
d = device.Device()
names, ms, sig = list(trees), {k: [] for k in trees}, {}
for rnd in range(rounds + 1):
    for k in (names if rnd % 2 == 0 else names[::-1]):
        d.upload_scene(dict(sc, bvh=trees[k])); d.resize(pr["width"], pr["height"])
        if rnd == 0:
            d.count_rays(True); d.reset_stats(); d.clear()
            d.render_frames(pr, [host.frame_seed(i) for i in range(2)]); d.sync()
            sig[k] = (int(d.stats().rays), hashlib.sha1(np.ascontiguousarray(d.read_accum()).view(np.uint8)).hexdigest()[:12])
            d.count_rays(False)
            continue
        d.render_frames(pr, [host.frame_seed(100 * rnd + i) for i in range(F)]); d.sync()
        d.render_frames(pr, [host.frame_seed(100 * rnd + 50 + i) for i in range(F)]); d.sync()
        ms[k].append(d.stats().kernel_ms_last / F)
base = float(np.median(ms[names[0]]))
print(f"{cfg}: {F} frames per launch, {rounds} rounds, alternated in one context")
for k in names:
    m = float(np.median(ms[k]))
    print(f"  {k:90s} {m:8.4f} ms/frame ({(m / base - 1) * 100:+5.2f} %)  rays {sig[k][0]}  image {sig[k][1]}")
print("images equal:", len({v[1] for v in sig.values()}) == 1)
