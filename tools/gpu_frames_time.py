"""Per-rank time per frame for world sizes 1,2,4,8 on ONE GPU (rank 0), with B frames per launch (glrtx_render_frames)."""
import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
if os.environ.get("GLRTX_LIB"):
    device.lib_path = lambda: device.LIB_DIR / os.environ["GLRTX_LIB"]
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc)
for world in ((1, 8) if os.environ.get("GLRTX_LIB") else (1, 2, 4, 8)):
    d.set_partition(0, world, 16); d.resize(1920, 1080)
    row = []
    for B in (1, 2, 4, 8, 16):
        ts = []
        for it in range(5):
            seeds = [host.frame_seed(it * B + f) for f in range(B)]
            d.render_frames(pr, seeds); d.sync(); ts.append(d.stats().kernel_ms_last)
        row.append(f"B{B} {np.median(ts[1:])/B:.3f}")
    print(os.environ.get("GLRTX_LIB", ""), "world", world, "ms per rank per frame:", "  ".join(row), flush=True)
