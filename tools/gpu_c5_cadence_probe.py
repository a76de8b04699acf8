"""Config 5, single-frame cadences per tree (CPU SAH | device SAH builder's), inside one process per tree: one glrtx_render per frame back to back with GLRTX_NO_FEED=1
(overlapped launches) and with a sync behind every call, wall time per frame; the launch's LDS and tree depth beside them."""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'opengl-raytracer_amd/python')
from glrt_amd import device, host, scenes
kind = sys.argv[1] if len(sys.argv) > 1 else "sah"
sc, pr = scenes.CONFIGS["c5"]()
d = device.Device()
if kind == "sah-gpu":
    d.build_bvh_sah(sc["vert"], sc["tri"])
    nodes, depth, ms = d.build_bvh_sah(sc["vert"], sc["tri"])
    sc = dict(sc, bvh=nodes)
    print("device tree: depth", depth, "build ms", ms)
d.upload_scene(sc); d.resize(pr["width"], pr["height"])
f = 0
def burst(n, sync_each):
    global f
    d.sync(); t0 = time.perf_counter()
    for k in range(n):
        d.render(dict(pr, seed=host.frame_seed(f))); f += 1
        if sync_each: d.sync()
    d.sync()
    return (time.perf_counter() - t0) * 1e3 / n
os.environ["GLRTX_NO_FEED"] = "1"
burst(48, False); burst(8, True)
for rep in range(3):
    a = burst(48, False); b = burst(32, True)
    st = d.stats()
    print(f"{kind}: overlapped {a:.4f}  synced {b:.4f} ms/frame   pipe_slots {st.pipe_slots} resident_max {st.pipe_resident_max} state {st.wf_state_mib} MiB", flush=True)
