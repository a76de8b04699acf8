"""Device LBVH build of config 5's 100k triangles (or N: argv[1]), a few times (for rocprofv3 --kernel-trace --stats: which kernels the build time is).
GLRTX_LIB=libglrtx_base.so (a file name under opengl-raytracer_amd/lib) times another build of the library; the nodes are compared with the CPU statement."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'opengl-raytracer_amd', 'python'))
import numpy as np
from glrt_amd import scenes, device, host
if os.environ.get("GLRTX_LIB"):
    device.lib_path = lambda: device.LIB_DIR / os.environ["GLRTX_LIB"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
sc, pr = scenes.config_c5(n=n, bvh="chain")
d = device.Device()
ms = [d.build_lbvh(sc["vert"], sc["tri"])[2] for _ in range(6)]
nodes, depth, _ = d.build_lbvh(sc["vert"], sc["tri"])
want, want_depth = host.build_bvh(sc["vert"], sc["tri"], "lbvh")
same = np.array_equal(np.asarray(nodes, np.float32).view(np.uint32).reshape(-1), np.asarray(want, np.float32).view(np.uint32).reshape(-1)) and depth == want_depth
print(os.environ.get("GLRTX_LIB", "libglrtx.so"), n, "triangles: device build ms:", [round(m, 3) for m in ms], "| equals the CPU statement bit for bit:", same)
