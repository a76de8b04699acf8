"""BASELINE config 3 (10k triangles, brute force = chain BVH): list scan vs the generic tree traversal."""
import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
sc, pr = scenes.config_c3()
imgs = {}
for mode in ("scan", "tree"):
    if mode == "tree": os.environ["GLRTX_NO_VINE_SCAN"] = "1"
    d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"]); d.count_rays(True)
    d.render(dict(pr, seed=host.frame_seed(0))); d.sync(); rays = d.stats().rays; imgs[mode] = d.read_accum(); d.count_rays(False)
    ts = []
    for it in range(3):
        d.render_frames(pr, [host.frame_seed(1 + 4 * it + f) for f in range(4)]); d.sync(); ts.append(d.stats().kernel_ms_last / 4)
    ms = float(np.median(ts))
    print(f"{mode}: {ms:.2f} ms/frame, {rays} rays x {sc['tri'].shape[0]} triangles = {rays*sc['tri'].shape[0]/ms/1e9:.2f} T triangle tests/s", flush=True)
    d.close()
print("bit-identical:", np.array_equal(imgs["scan"].view(np.uint32), imgs["tree"].view(np.uint32)))
