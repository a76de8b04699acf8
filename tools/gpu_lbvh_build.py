"""Device LBVH build of config 5's 100k triangles, a few times (for rocprofv3 --kernel-trace --stats: which kernels the build time is)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'opengl-raytracer_amd', 'python'))
from glrt_amd import scenes, device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
sc, pr = scenes.config_c5(n=n, bvh="chain")
d = device.Device()
ms = [d.build_lbvh(sc["vert"], sc["tri"])[2] for _ in range(6)]
print("device build ms:", [round(m, 3) for m in ms])
