"""A/B timing of kernel variants in ONE process (interleaved rounds), plus bit-equality between them."""
import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sc, pr = scenes.CONFIGS[cfg]()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
imgs = {}
for v in (0, 1, 2):
    d.set_variant(v); d.clear(); d.reset_stats(); d.count_rays(True)
    d.render(dict(pr, seed=host.frame_seed(0))); d.sync()
    imgs[v] = d.read_accum(); print("variant", v, "rays", d.stats().rays, flush=True)
print("bit-identical 0/1:", np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32)), "0/2:", np.array_equal(imgs[0].view(np.uint32), imgs[2].view(np.uint32)), "mismatching px:", int((imgs[0].view(np.uint32) != imgs[2].view(np.uint32)).any(-1).sum()))
d.count_rays(False)
res = {0: [], 1: [], 2: []}
for r in range(rounds):
    for v in (0, 1, 2):
        d.set_variant(v)
        for f in range(3):
            d.render(dict(pr, seed=host.frame_seed(f + 1))); d.sync(); res[v].append(d.stats().kernel_ms_last)
for v in (0, 1, 2):
    a = np.array(res[v]); print(f"variant {v}: median {np.median(a):.3f} ms min {a.min():.3f} max {a.max():.3f}")
