"""Experiment (round 5): the visiting order at the forks above the LIGHTS.  The reference visits children.y first (raytrace.frag:299-307) whatever the ray; the tree -- and so
the order of a fork's two children -- is the builder's choice.  A shadow ray ends its search as soon as the light is hit (everything beyond is culled by tHit from then on), so a
tree whose forks put the child that contains light triangles in the y slot lets every shadow ray find its light first.  This tool swaps the children of such forks in the CPU SAH
tree and renders both trees alternately in one context.
    python tools/gpu_light_first.py [config] [frames per launch] [rounds]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
import numpy as np
from glrt_amd import device, host, scenes

cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 10
sc, pr = scenes.CONFIGS[cfg]()


def light_first(nodes, tri, mat):
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
    swapped = 0
    for i in order:
        if N[i, 8] < 0:
            x, y = int(N[i, 6]), int(N[i, 7])
            if has[x] and not has[y]:
                N[i, 6], N[i, 7] = y, x
                swapped += 1
    return N.reshape(-1, 3), swapped


raw = sc.get("bvh_builder", sc["bvh"])  # the builder's own child order (scenes.SceneBuilder.build applies the pass since round 5: scene["bvh"])
lf, swapped = light_first(raw, sc["tri"], sc["mat"])
assert np.array_equal(np.asarray(lf, np.float32).reshape(-1, 3), np.asarray(host.lights_first(raw, sc["tri"], sc["mat"])[0])), "this tool's statement of the pass == glrt_bvh_lights_first"
trees = {"cpu binned SAH, the builder's child order": raw, f"... with the light side first at {swapped} forks": lf}
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
    print(f"  {k:60s} {m:8.4f} ms/frame ({(m / base - 1) * 100:+5.2f} %)  rays {sig[k][0]}  image {sig[k][1]}")
print("images equal:", len({v[1] for v in sig.values()}) == 1)
