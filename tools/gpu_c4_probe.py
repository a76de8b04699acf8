"""Config 4 (4K, 16 spp), eight glrtx_render calls back to back on the context's own stream, twice: issue time, wall time per frame, kernel launches, fed launches and
appended frames -- does a burst behind a PLAIN launch (a frame whose sample planes exceed the overlapped form's 1 GiB) become a fed launch?  (profiles/r06_launch_shapes.txt)
    python tools/gpu_c4_probe.py"""
import sys, os, time
sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
from glrt_amd import device, host, scenes
sc, pr = scenes.CONFIGS["c4"]()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
d.use_own_stream(True) if hasattr(d, "use_own_stream") else None
for rep in range(2):
    d.sync(); s0 = d.stats(); t = time.perf_counter()
    for i in range(8):
        d.render(dict(pr, seed=host.frame_seed(rep * 8 + i)))
    t1 = time.perf_counter(); d.sync(); t2 = time.perf_counter(); s1 = d.stats()
    print("rep", rep, "issue ms", (t1 - t) * 1e3, "total ms/frame", (t2 - t) * 1e3 / 8, "kernel launches", s1.kernel_launches - s0.kernel_launches, "feed launches", s1.feed_launches - s0.feed_launches,
          "appended", s1.feed_appended - s0.feed_appended, "pipe_slots", s1.pipe_slots, "frames_last", s1.frames_last, flush=True)
