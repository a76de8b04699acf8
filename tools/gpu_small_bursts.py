"""Small multi-frame calls on the context's own stream (fed launches): ONE glrtx_render_frames(n) call on an idle device, and 24 such calls back to back, per library.
    python tools/gpu_small_bursts.py LIB [spp = 0 (the config's)] [config = headline]
Kernel time per frame of the lone call; wall time per frame of the burst."""
import os, pathlib, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
from glrt_amd import device, host, scenes  # noqa: E402
lib, spp, config = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0, sys.argv[3] if len(sys.argv) > 3 else "headline"
device.lib_path = lambda: pathlib.Path(os.path.join(ROOT, lib))
sc, pr = scenes.CONFIGS[config]()
if spp: pr = dict(pr, n_samples=spp)
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
f = 0
def seeds(n):
    global f
    s = [host.frame_seed(f + i) for i in range(n)]; f += n
    return s
for n in (2, 3, 4, 6):
    lone, burst = [], []
    for rep in range(4):
        d.sync(); d.render_frames(pr, seeds(n)); d.sync()
        lone.append(d.stats().kernel_ms_last / d.stats().frames_last)
        d.sync(); t0 = time.perf_counter()
        for k in range(24): d.render_frames(pr, seeds(n))
        d.sync(); burst.append((time.perf_counter() - t0) * 1e3 / (24 * n))
    print(f"{os.path.basename(lib):22s} {config} {pr['n_samples']:2d} spp  render_frames({n}): lone {min(lone[1:]):8.4f}   24 calls back to back {min(burst[1:]):8.4f} ms/frame", flush=True)
