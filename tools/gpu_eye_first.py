"""Experiment (round 5): after the light side (tools/gpu_light_first.py), the EYE side: at the forks that have no one-sided lights, the child whose box lies nearer to the camera
in the y slot (front-to-back for the primary rays, a fifth of all rays).  Variants rendered alternately in one context.
    python tools/gpu_eye_first.py [config] [frames per launch] [rounds]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
import numpy as np
from glrt_amd import device, host, scenes

cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 10
sc, pr = scenes.CONFIGS[cfg]()
eye = np.asarray(pr["c2w"], np.float32).reshape(4, 4)[3, :3]  # column-major: the translation column


def eye_first(nodes, tri, mat, mode):
    N = np.array(nodes, np.float32).reshape(-1, 9).copy()
    emissive = np.linalg.norm(np.asarray(mat, np.float32).reshape(-1, 6, 3)[:, 1], axis=1) != 0
    is_light_tri = emissive[np.asarray(tri, np.float32).reshape(-1, 4)[:, 3].astype(int)]
    has = np.zeros(N.shape[0], bool)
    order, st = [], [0]
    while st:
        i = st.pop(); order.append(i)
        if N[i, 8] < 0: st += [int(N[i, 6]), int(N[i, 7])]
    for i in reversed(order):
        has[i] = is_light_tri[int(N[i, 8])] if N[i, 8] >= 0 else (has[int(N[i, 6])] or has[int(N[i, 7])])
    def dist(i):  # distance from the eye to the box
        lo, hi = N[i, 0:3], N[i, 3:6]
        d = np.maximum(np.maximum(lo - eye, eye - hi), 0.0)
        return float(np.dot(d, d))
    def cdist(i):
        c = 0.5 * (N[i, 0:3] + N[i, 3:6]) - eye
        return float(np.dot(c, c))
    swapped = 0
    for i in order:
        if N[i, 8] >= 0: continue
        x, y = int(N[i, 6]), int(N[i, 7])
        if has[x] != has[y]: continue  # the light side decides
        dx, dy = (dist(x), dist(y)) if mode == "box" else (cdist(x), cdist(y))
        if dx < dy or (dx == dy and cdist(x) < cdist(y)):
            N[i, 6], N[i, 7] = y, x; swapped += 1
    return N.reshape(-1, 3), swapped


trees = {"light side first (current)": sc["bvh"]}
for mode in ("box", "centre"):
    t, n = eye_first(sc["bvh"], sc["tri"], sc["mat"], mode)
    trees[f"+ eye side first by {mode} distance ({n} forks)"] = t
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
print(f"{cfg}: {F} frames per launch, {rounds} rounds, alternated in one context; eye {eye}")
for k in names:
    m = float(np.median(ms[k]))
    print(f"  {k:60s} {m:8.4f} ms/frame ({(m / base - 1) * 100:+5.2f} %)  rays {sig[k][0]}  image {sig[k][1]}")
print("images equal:", len({v[1] for v in sig.values()}) == 1)
