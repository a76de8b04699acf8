"""Experiment (round 5): spatial splits of LARGE triangles ("early split clipping", Ernst & Greiner 2007) in front of the SAH builder.  The wire format has one triangle index per
LEAF, not one leaf per triangle: a big triangle -- a wall of the box, whose own bounds cover the room -- may be referenced by several leaves, each under forks whose boxes bound
only the piece of it that lies there (the triangle clipped to a cell, bounds rounded outwards).  The reference's traversal tests the triangle wherever a fork's box lets the ray
in; a second test of the same triangle finds the same t and changes nothing (strict <).  This tool builds such trees in Python around the existing SAH builder (every piece is
handed to it as a degenerate triangle spanning the piece's box; the leaves are then pointed back at the original triangles), renders them next to the plain tree in ONE context
and prints time, ray count and image checksum.
    python tools/gpu_split_study.py [config] [frames per launch] [rounds]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
import numpy as np
from glrt_amd import device, host, scenes

cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sc, pr = scenes.CONFIGS[cfg]()
V = np.asarray(sc["vert"], np.float32).reshape(-1, 5, 3)[:, 0].astype(np.float64)
T = np.asarray(sc["tri"], np.float32).reshape(-1, 4)


def clip(poly, axis, value, keep_below):
    out = []
    n = len(poly)
    for i in range(n):
        a, b = poly[i], poly[(i + 1) % n]
        ia = (a[axis] <= value) if keep_below else (a[axis] >= value)
        ib = (b[axis] <= value) if keep_below else (b[axis] >= value)
        if ia: out.append(a)
        if ia != ib:
            t = (value - a[axis]) / (b[axis] - a[axis])
            p = a + t * (b - a); p[axis] = value
            out.append(p)
    return out


def split_refs(max_extent):
    refs = []  # (triangle, lo, hi)
    for t in range(T.shape[0]):
        work = [[V[int(T[t, k])].copy() for k in range(3)]]
        while work:
            poly = work.pop()
            P = np.array(poly)
            lo, hi = P.min(0), P.max(0)
            ext = hi - lo
            ax = int(np.argmax(ext))
            if ext[ax] <= max_extent or len(poly) < 3:
                refs.append((t, np.nextafter(lo.astype(np.float32), np.float32(-np.inf)), np.nextafter(hi.astype(np.float32), np.float32(np.inf))))
                continue
            mid = 0.5 * (lo[ax] + hi[ax])
            a, b = clip(poly, ax, mid, True), clip(poly, ax, mid, False)
            if len(a) >= 3: work.append(a)
            if len(b) >= 3: work.append(b)
    return refs


def tree_over(refs):
    n = len(refs)
    fv = np.zeros((n * 3, 5, 3), np.float32)
    for i, (t, lo, hi) in enumerate(refs):
        fv[3 * i, 0] = lo; fv[3 * i + 1, 0] = hi; fv[3 * i + 2, 0] = lo
    ft = np.concatenate([np.arange(n * 3, dtype=np.float32).reshape(n, 3), np.zeros((n, 1), np.float32)], 1)
    nodes, depth = host.build_bvh(fv.reshape(-1, 3), ft, "sah")
    N = np.asarray(nodes, np.float32).reshape(-1, 9).copy()
    leaf = N[:, 8] >= 0
    N[leaf, 8] = np.array([refs[int(r)][0] for r in N[leaf, 8]], np.float32)
    N3, _ = host.lights_first(N.reshape(-1, 3), sc["tri"], sc["mat"])
    return N3, depth


ext = (V.max(0) - V.min(0)).max()
trees = {"plain SAH tree (light side first)": (sc["bvh"], sc["bvh_depth"], T.shape[0])}
for div in (4, 8, 16, 32):
    refs = split_refs(ext / div)
    nodes, depth = tree_over(refs)
    trees[f"triangles split to <= extent / {div}: {len(refs)} references"] = (nodes, depth, len(refs))
d = device.Device()
names, ms, sig = list(trees), {k: [] for k in trees}, {}
for rnd in range(rounds + 1):
    for k in (names if rnd % 2 == 0 else names[::-1]):
        nodes, depth, _ = trees[k]
        d.upload_scene(dict(sc, bvh=nodes, bvh_depth=depth)); d.resize(pr["width"], pr["height"])
        if rnd == 0:
            d.count_rays(True); d.reset_stats(); d.clear()
            d.render_frames(pr, [host.frame_seed(i) for i in range(2)]); d.sync()
            sig[k] = (int(d.stats().rays), hashlib.sha1(np.ascontiguousarray(d.read_accum()).view(np.uint8)).hexdigest()[:12], int(d.stats().stack_entries), int(d.stats().n_fork))
            d.count_rays(False)
            continue
        d.render_frames(pr, [host.frame_seed(100 * rnd + i) for i in range(F)]); d.sync()
        d.render_frames(pr, [host.frame_seed(100 * rnd + 50 + i) for i in range(F)]); d.sync()
        ms[k].append(d.stats().kernel_ms_last / F)
base = float(np.median(ms[names[0]]))
print(f"{cfg}: {T.shape[0]} triangles, scene extent {ext:g}; {F} frames per launch, {rounds} rounds, alternated in one context")
for k in names:
    m = float(np.median(ms[k]))
    print(f"  {k:62s} {m:8.4f} ms/frame ({(m / base - 1) * 100:+6.2f} %)  rays {sig[k][0]}  image {sig[k][1]}  stack {sig[k][2]}  forks {sig[k][3]}")
print("images equal:", len({v[1] for v in sig.values()}) == 1)
