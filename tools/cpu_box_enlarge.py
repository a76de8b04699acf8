"""Does the reference's image depend on the exact boxes of the tree?  The oracle (CPU restatement, pinned to the reference's shader) renders one frame of a config with its tree and
with every box of that tree enlarged -- by one ulp, by 1e-5 relative, by 1e-3 absolute -- and counts the pixels that differ.  (CPU only; test infrastructure, like everything that
uses oracle/.)  A conservative box can only ADD triangle tests; a pixel changes where the reference's float slab test (raytrace.frag:259-274) rejects a box whose triangle the
ray would hit -- flat boxes of axis-aligned faces, grazing rays.  -> profiles/r06_box_enlarge.txt"""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/opengl-raytracer_amd/python')
from glrt_amd import scenes, host
from oracle import pt_oracle
for name, kw in (("headline", dict(width=960, height=540)), ("c2", dict(width=960, height=540)), ("c5", dict(width=480, height=270))):
    sc, pr = scenes.CONFIGS[name](**kw)
    a, ra = pt_oracle.render(sc, pr)
    bvh = np.array(sc["bvh"], dtype=np.float32, copy=True).reshape(-1, 9)
    for label, grow in (("1 ulp", None), ("1e-5 rel", 1e-5), ("1e-3 abs", -1e-3)):
        b = bvh.copy()
        if grow is None:
            b[:, 0:3] = np.nextafter(b[:, 0:3], np.float32(-np.inf)); b[:, 3:6] = np.nextafter(b[:, 3:6], np.float32(np.inf))
        elif grow > 0:
            ext = np.maximum(np.abs(b[:, 0:3]), np.abs(b[:, 3:6])) * np.float32(grow)
            b[:, 0:3] -= ext; b[:, 3:6] += ext
        else:
            b[:, 0:3] -= np.float32(-grow); b[:, 3:6] += np.float32(-grow)
        sc2 = dict(sc, bvh=b.reshape(np.array(sc["bvh"]).shape))
        c, rc = pt_oracle.render(sc2, pr)
        diff = (a.view(np.uint32) != c.view(np.uint32)).any(axis=-1)
        print(f"{name}: boxes enlarged by {label}: {int(diff.sum())} of {diff.size} pixels differ, rays {ra} -> {rc}", flush=True)
