"""One launch per frame, back to back (the reference's cadence, window.cpp:121-169): wall time per frame of N unthrottled glrtx_render calls
followed by one sync, for every BASELINE config at full size.  usage: python tools/gpu_cadence.py [frames ...]   (default 20 100)"""
import sys, os, time; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import torch; torch.cuda.init()
from glrt_amd import scenes, device, host
counts = [int(x) for x in sys.argv[1:]] or [20, 100]
d = device.Device()
only = os.environ.get("CADENCE_CONFIGS", "").split(",") if os.environ.get("CADENCE_CONFIGS") else None
for name, kw in (("c1", {}), ("c2", {}), ("c3", dict(bvh="sah")), ("c4", dict(n_samples=1)), ("c5", {}), ("headline", {})):
    if only and name not in only: continue
    sc, pr = scenes.CONFIGS[name](**kw)
    d.upload_scene(sc); d.set_partition(0, 1, 16); d.resize(pr["width"], pr["height"])
    out = []
    f0 = 0
    for n in [8] + counts:  # (8: warm-up, buffers allocated)
        d.sync(); t0 = time.perf_counter()
        for f in range(n): d.render(dict(pr, seed=host.frame_seed(f0 + f)))
        d.sync(); dt = (time.perf_counter() - t0) / n * 1e3; f0 += n
        out.append(f"{n} frames: {dt:.3f} ms/frame")
    print(f"{name} {kw} {pr['width']}x{pr['height']} depth {pr['max_depth']}: " + "   ".join(out[1:]), flush=True)
