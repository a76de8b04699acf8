"""What a better tree would buy: one scene, several trees, rendered ALTERNATELY inside one context (the context-to-context spread is larger than the
differences of interest).  Trees: the CPU binned-SAH tree, the device-built LBVH, and the CPU statement of the LBVH builder with the exact-sweep SAH rebuild applied to subtrees
of up to N leaves (host libraries compiled with -DGLRT_LBVH_REBUILD_LEAVES=N under opengl-raytracer_amd/lib/study/; N = 1000000: the whole tree is one exact-sweep SAH tree).
Prints ms per frame (median of the rounds), traversal steps are in tools/gpu_travstats.py.

    python tools/gpu_tree_study.py [config] [frames per launch] [rounds]"""
import ctypes as C
import os
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "opengl-raytracer_amd" / "python"))
from glrt_amd import device, host, scenes  # noqa: E402

STATS = "--stats" in sys.argv  # second mode: the traversal-statistics build (libglrtx_stats.so, `make diag`): steps / fork visits / triangle tests per ray of every tree
if STATS:
    sys.argv.remove("--stats")
    device.lib_path = lambda: device.LIB_DIR / "libglrtx_stats.so"
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sc, pr = scenes.CONFIGS[cfg]()
d = device.Device()
trees = {"cpu binned SAH": (sc["bvh"], sc.get("bvh_depth", 0), 0.0)}
nodes, depth, ms = d.build_lbvh(sc["vert"], sc["tri"])
trees["device LBVH (rebuild <= 64)"] = (nodes, depth, ms)
t = time.perf_counter()
sl_nodes, sl_depth = host.build_bvh(sc["vert"], sc["tri"], "sahl")
trees["cpu SAH by levels (+ exact sweep <= 64)"] = (sl_nodes, sl_depth, (time.perf_counter() - t) * 1e3)
if hasattr(d, "build_bvh_sah"):
    nodes, depth, ms = d.build_bvh_sah(sc["vert"], sc["tri"])
    trees["device SAH by levels"] = (nodes, depth, ms)
vert = np.ascontiguousarray(sc["vert"], np.float32).reshape(-1, 15)
tri = np.ascontiguousarray(sc["tri"], np.float32).reshape(-1, 4)
fp = C.POINTER(C.c_float)
for lib in sorted((ROOT / "opengl-raytracer_amd" / "lib" / "study").glob("libglrt_host_rl*.so"), key=lambda p: int(p.stem.split("rl")[1])):
    L = C.CDLL(str(lib))
    L.glrt_bvh_build_lbvh.argtypes = [fp, C.c_size_t, fp, C.c_size_t, fp, C.POINTER(C.c_int)]
    out = np.zeros(((2 * tri.shape[0] - 1) * 3, 3), np.float32)
    dep = C.c_int(0)
    t = time.perf_counter()
    rc = L.glrt_bvh_build_lbvh(vert.ctypes.data_as(fp), vert.shape[0], tri.ctypes.data_as(fp), tri.shape[0], out.ctypes.data_as(fp), C.byref(dep))
    assert rc == 0, rc
    trees[f"cpu LBVH, rebuild <= {lib.stem.split('rl')[1]}"] = (out, int(dep.value), (time.perf_counter() - t) * 1e3)
if STATS:
    L = device.lib()
    print(f"{cfg}: traversal statistics per tree (2 frames; steps = lane-steps of the traversal kernel, rays = every intersect() of the reference)")
    for k, (nodes, depth, _) in trees.items():
        d.upload_scene(dict(sc, bvh=nodes, bvh_depth=depth)); d.resize(pr["width"], pr["height"]); d.count_rays(True); d.reset_stats()
        o0 = (C.c_ulonglong * 8)(); L.glrtx_debug_trav_stats(o0); o0 = list(o0)
        d.render_frames(pr, [host.frame_seed(i) for i in range(2)]); d.sync()
        o1 = (C.c_ulonglong * 8)(); L.glrtx_debug_trav_stats(o1); o = [b - a for a, b in zip(o0, list(o1))]
        rays = int(d.stats().rays)
        st = d.stats()
        print(f"  {k:42s} steps/ray {o[1] / rays:6.2f}  fork visits/ray {o[2] / rays:6.2f}  triangle tests/ray {o[3] / rays:5.2f}  lanes/wave-step {o[1] / max(o[0], 1):5.1f}  "
              f"records/wave-step {o[5] / max(o[0], 1):5.1f}  forks in the packed tree {st.n_fork}  stack {st.stack_entries}")
    sys.exit(0)
ms = {k: [] for k in trees}
sig = {}
names = list(trees)
for rnd in range(rounds + 1):
    for k in (names if rnd % 2 == 0 else names[::-1]):
        nodes, depth, _ = trees[k]
        d.upload_scene(dict(sc, bvh=nodes, bvh_depth=depth)); d.resize(pr["width"], pr["height"])
        if rnd == 0:
            d.count_rays(True); d.reset_stats(); d.clear()
            d.render_frames(pr, [host.frame_seed(i) for i in range(2)]); d.sync()
            import hashlib
            sig[k] = (int(d.stats().rays), hashlib.sha1(np.ascontiguousarray(d.read_accum()).view(np.uint8)).hexdigest()[:12], int(d.stats().stack_entries))
            d.count_rays(False)
            continue
        d.render_frames(pr, [host.frame_seed(100 * rnd + i) for i in range(F)]); d.sync()
        d.render_frames(pr, [host.frame_seed(100 * rnd + 50 + i) for i in range(F)]); d.sync()
        ms[k].append(d.stats().kernel_ms_last / F)
base = float(np.median(ms[names[0]]))
print(f"{cfg}: {tri.shape[0]} triangles, {F} frames per launch, {rounds} rounds, trees alternated inside one context")
for k in names:
    m = float(np.median(ms[k]))
    print(f"  {k:42s} {m:8.4f} ms/frame ({(m / base - 1) * 100:+5.2f} %)  build {trees[k][2]:9.2f} ms  depth {trees[k][1]:3d}  stack {sig[k][2]:2d}  rays {sig[k][0]}  image {sig[k][1]}")
print("images equal:", len({v[1] for v in sig.values()}) == 1)
