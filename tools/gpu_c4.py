import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
sc, pr = scenes.config_c4(n_samples=1)
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
for B in (16, 64):
    d.reset_stats()
    d.render_frames(pr, [host.frame_seed(f) for f in range(B)]); d.sync()
    st = d.stats()
    print(f"c4 4K B={B}: {st.kernel_ms_total/B:.3f} ms/frame, kernel launches {st.kernel_launches}, frames {st.launches}", flush=True)
sc, pr = scenes.config_c4(n_samples=16)
d.clear(); d.reset_stats()
d.render_frames(pr, [host.frame_seed(f) for f in range(4)]); d.sync()
st = d.stats()
print(f"c4 4K 16 spp x 4 frames in flight: {st.kernel_ms_total/4:.3f} ms/frame ({st.kernel_ms_total/64:.3f} per sample), launches {st.kernel_launches}; count==64: {bool(np.all(d.read_accum()[...,3]==64.0))}", flush=True)
