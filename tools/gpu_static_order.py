"""Experiment (round 6): a STATIC order of every fork's children -- no calibration frame -- by box area per unit cost (area / n^e, n = triangles under the child): the child with the
higher score into the slot the traversal visits first, or the lower one ("inv"); the light side first applied on top, as the scene builders do.  One context, trees alternating.

    python tools/gpu_static_order.py [configs ...]"""
import hashlib
import sys

import numpy as np

sys.path.insert(0, "."); sys.path.insert(0, "opengl-raytracer_amd/python")
from glrt_amd import device, host, scenes  # noqa: E402


def static_order(nodes, e, inverse):
    N = np.array(nodes, np.float32).reshape(-1, 9).copy()
    cnt = np.zeros(N.shape[0], np.int64)
    order, st = [], [0]
    while st:
        i = st.pop(); order.append(i)
        if N[i, 8] < 0:
            st += [int(c) for c in N[i, 6:8] if c >= 0]
    for i in reversed(order):
        cnt[i] = 1 if N[i, 8] >= 0 else sum(cnt[int(c)] for c in N[i, 6:8] if c >= 0)
    d = np.maximum(N[:, 3:6] - N[:, 0:3], 0).astype(np.float64)
    area = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]
    sw = 0
    for i in order:
        if N[i, 8] < 0 and N[i, 6] >= 0 and N[i, 7] >= 0:
            x, y = int(N[i, 6]), int(N[i, 7])
            sx, sy = area[x] / cnt[x] ** e, area[y] / cnt[y] ** e
            if (sx > sy) != inverse and sx != sy:
                N[i, 6], N[i, 7] = y, x; sw += 1
    return N.reshape(-1, 3), sw


cfgs = sys.argv[1:] or ["headline", "c2", "c4", "c5"]
d = device.Device()
for name in cfgs:
    sc, pr = scenes.CONFIGS[name]() if name != "c4" else scenes.CONFIGS[name](n_samples=1)
    W, H = pr["width"], pr["height"]
    trees = {"default": (sc["bvh"], 0)}
    for e in (0.0, 0.5, 1.0):
        for inv in (False, True):
            nodes, sw = static_order(sc["bvh_builder"], e, inv)
            nodes, lf = host.lights_first(nodes, sc["tri"], sc["mat"])
            trees[f"area/n^{e:g}{' inv' if inv else ''}"] = (nodes, sw)
    F, rounds = 16, 7
    ms = {k: [] for k in trees}
    names = list(trees)
    for rnd in range(rounds + 1):
        for k in (names if rnd % 2 == 0 else names[::-1]):
            d.upload_scene(dict(sc, bvh=trees[k][0])); d.resize(W, H); d.clear()
            d.render_frames(pr, [host.frame_seed(100 * rnd + i) for i in range(F)]); d.sync()
            if rnd: ms[k].append(d.stats().kernel_ms_last / F)
    base = float(np.median(ms["default"]))
    print(f"== {name}, {F} frames per launch, {rounds} rounds", flush=True)
    for k in names:
        m = float(np.median(ms[k]))
        print(f"  {k:18s} {m:.4f} ms/frame ({(m / base - 1) * 100:+.2f} %)  forks exchanged {trees[k][1]}", flush=True)
