"""Diagnostic (GPU box, lib built with -DGLRTX_RAY_LOG: make -C opengl-raytracer_amd diag): record the ray queues of one launch of the render kernel,
then run the traverse phase ALONE over them (pt_replay_traverse): same rays, same grouping into workgroup trips, no path state, no shade phase.
Compares what a ray costs in the traverse phase in situ with what it costs when nothing else streams through the caches; under
`rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum` the replay kernel's counters are the node fetches' L2 hit rate by themselves (plus the 32-byte ray records).
Usage: gpu_replay.py [config] [frames per launch] [replays]"""
import os, sys, ctypes as C
os.environ.setdefault("GLRTX_SUSPEND_MAX", "0")  # every ray of a trip is in its queue (a parked ray would be resumed from the suspend area)
sys.path.insert(0, '.'); sys.path.insert(0, 'opengl-raytracer_amd/python')
from glrt_amd import scenes, device, host
device.lib_path = lambda: device.LIB_DIR / os.environ.get("GLRTX_REPLAY_LIB", "libglrtx_raylog.so")
cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
sc, pr = scenes.CONFIGS[cfg]()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
L = device.lib()
L.glrtx_debug_ray_log_begin.argtypes = [C.c_void_p, C.c_ulonglong, C.c_uint]
L.glrtx_debug_ray_log_replay.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
seeds = [host.frame_seed(f) for f in range(B)]
d.render_frames(pr, seeds); d.sync()                       # warm (buffers exist)
d.reset_stats(); d.render_frames(pr, seeds); d.sync()
ms_plain = d.stats().kernel_ms_last
cap = int(13e6 * B)
assert L.glrtx_debug_ray_log_begin(d.h, cap, 1 << 20) == 0
d.reset_stats(); d.render_frames(pr, seeds); d.sync()       # recorded
ms_rec = d.stats().kernel_ms_last
out = (C.c_double * 4)()
rc = L.glrtx_debug_ray_log_replay(d.h, reps, out)
assert rc == 0, rc
ms, logged, trips, offered = out[0], out[1], out[2], out[3]
print(f"{cfg}: {B} frames per launch; render kernel {ms_plain:.3f} ms ({ms_plain / B:.3f} per frame; {ms_rec:.3f} ms while recording)")
print(f"log: {int(logged)} ray records in {int(trips)} workgroup trips ({int(offered)} offered); per frame {logged / B / 1e6:.2f} M records")
print(f"replay of the traverse phase alone: {ms:.3f} ms = {ms / B:.3f} ms per frame = {100.0 * ms / ms_plain:.1f} % of the render kernel's time (in situ the phase is 69 % of a workgroup's time)")

# ---- reordering experiments: the same rays, the same trips, another order inside each trip's queue (what would a sort before the phase buy?)
if os.environ.get("GLRTX_REPLAY_ORDERS", "1") != "0":
    import numpy as np
    L.glrtx_debug_ray_log_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_void_p, C.c_uint, C.c_int]
    n_rec, n_tr = int(offered), int(trips)
    n_rec = min(n_rec, cap)
    rays = np.zeros((n_rec, 8), np.float32)
    tr = np.zeros((n_tr, 2), np.uint32)
    assert L.glrtx_debug_ray_log_copy(d.h, rays.ctypes.data, n_rec, tr.ctypes.data, n_tr, 0) == 0
    # trip index of every record (records outside any logged trip keep -1 and stay where they are)
    trip_of = np.full(n_rec, -1, np.int64)
    order0 = np.argsort(tr[:, 0], kind="stable")
    for t in order0:
        o, n = int(tr[t, 0]), int(tr[t, 1])
        trip_of[o:o + n] = t
    rid = rays[:, 3].view(np.uint32)
    valid = rid != 0xFFFFFFFF
    shadow = (rid & 1) == 1
    lo = rays[valid, 0:3].min(axis=0); hi = rays[valid, 0:3].max(axis=0)
    span = np.maximum(hi - lo, 1e-6)

    def morton(cells, bits):
        q = np.clip(((rays[:, 0:3] - lo) / span * cells).astype(np.int64), 0, cells - 1)
        key = np.zeros(n_rec, np.int64)
        for b in range(bits):
            for a in range(3):
                key |= ((q[:, a] >> b) & 1) << (3 * b + a)
        return key
    octant = ((rays[:, 4] < 0).astype(np.int64) | ((rays[:, 5] < 0).astype(np.int64) << 1) | ((rays[:, 6] < 0).astype(np.int64) << 2))
    rng = np.random.default_rng(1)
    keys = {
        "as recorded": np.arange(n_rec, dtype=np.int64),
        "shuffled inside each trip": rng.permutation(n_rec).astype(np.int64),
        "origin cell 16^3 (Morton), then direction octant": morton(16, 4) * 8 + octant,
        "origin cell 64^3 (Morton), then direction octant": morton(64, 6) * 8 + octant,
        "direction octant, then origin cell 16^3": octant * (1 << 12) + morton(16, 4),
        "direction octant, then origin cell 64^3": octant * (1 << 18) + morton(64, 6),
    }
    # the two rays of one bounce -- the path's next ray and its shadow ray -- leave from the same point: put them next to each other (groups in recorded order)
    ob = np.ascontiguousarray(rays[:, 0:3]).view(np.uint32).astype(np.uint64)
    okey = (ob[:, 0] * np.uint64(0x9E3779B97F4A7C15) ^ ob[:, 1] * np.uint64(0xC2B2AE3D27D4EB4F) ^ ob[:, 2] * np.uint64(0x165667B19E3779F9))
    _, first, inv = np.unique(okey, return_index=True, return_inverse=True)
    keys["rays of one origin adjacent (a path's next ray and its shadow ray), groups as recorded"] = first[inv].astype(np.int64)
    if os.environ.get("GLRTX_REPLAY_ONLY_ORIGIN"):
        keys = {k: v for k, v in keys.items() if k.startswith("as recorded") or k.startswith("rays of one origin")}
    base_ms = None
    for name, key in keys.items():
        # stable sort by (trip, key); invalid records (skip markers) keep their relative place at the end of their trip
        k2 = np.where(valid, key, np.iinfo(np.int64).max)
        perm = np.lexsort((k2, trip_of))
        # records with trip -1 sort first and are mapped onto themselves; the others fill their trips' slots in order
        dst = np.lexsort((np.arange(n_rec), trip_of))  # positions grouped by trip, in address order
        new = rays.copy()
        new[dst] = rays[perm]
        assert L.glrtx_debug_ray_log_copy(d.h, new.ctypes.data, n_rec, None, 0, 1) == 0
        assert L.glrtx_debug_ray_log_replay(d.h, reps, out) == 0
        if base_ms is None:
            base_ms = out[0]
        print(f"  order: {name:52s} {out[0]:8.3f} ms ({100.0 * out[0] / base_ms:6.1f} % of as recorded)")
    print("(shadow rays carry their path's queue position, path rays their path id: both are independent of the order, so every order does the same work)")
