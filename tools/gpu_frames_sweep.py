"""Kernel time per frame of ONE glrtx_render_frames launch against the number of frames in it and the paths a workgroup keeps alive (device idle before and after).

    python tools/gpu_frames_sweep.py [config] [n,n,n ...] [spp] [block_paths,block_paths ... (0 = the library's choice)]

Says where a config's per-frame time comes from at small launches: a launch ends with a tail in which the last paths run out -- about one path item's lifetime, and an item
is a pixel's n_samples samples in sequence (their random numbers are one chain) -- so a launch that covers few helpings of long items spends much of its time in the tail."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
from glrt_amd import device, host, scenes  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "c4"
counts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8,16").split(",")]
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
blocks = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0").split(",")]
sc, pr = scenes.CONFIGS[config]()
if spp: pr = dict(pr, n_samples=spp)
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
f0 = 0
print(f"{config} {pr['width']}x{pr['height']} {pr['n_samples']} spp: ms per frame (kernel time of the call's last launch / its frames)")
print("frames/launch " + " ".join(f"{('bp ' + str(b)) if b else 'default':>10s}" for b in blocks))
for n in counts:
    row = []
    for b in blocks:
        if b: os.environ["GLRTX_BLOCK_PATHS"] = str(b)
        else: os.environ.pop("GLRTX_BLOCK_PATHS", None)
        best = None
        for rep in range(3):
            seeds = [host.frame_seed(f0 + i) for i in range(n)]; f0 += n
            d.sync()
            d.render_frames(pr, seeds) if n > 1 else d.render(dict(pr, seed=seeds[0]))
            d.sync()
            st = d.stats()
            if rep: best = min(best or 1e9, st.kernel_ms_last / st.frames_last)
        row.append(best)
    print(f"{n:13d} " + " ".join(f"{x:10.4f}" for x in row), flush=True)
