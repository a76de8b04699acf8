import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device
from oracle import pt_oracle
def cmp(name, sc, pr):
    t=time.time(); acc_o, rays_o = pt_oracle.render(sc, pr); to=time.time()-t
    acc_g, rays_g, ms = device.render_image(sc, pr)
    bit = (acc_o.view(np.uint32) == acc_g.view(np.uint32)).all(-1)
    d = np.abs(acc_o-acc_g)
    print(f"{name}: oracle {to:.2f}s gpu {ms:.3f}ms rays {rays_o} vs {rays_g} bitexact {bit.mean():.6f} mismatch {(~bit).sum()} maxabs {np.nanmax(d):.3e} within1e-4 {(d.max(-1)<=1e-4).mean():.6f}", flush=True)
sc, pr = scenes.config_c1(256,256,max_depth=1); cmp("c1 256 d1", sc, pr)
sc, pr = scenes.config_c1(200,120,max_depth=16, n_samples=4); cmp("c1 200x120 d16 spp4", sc, pr)
pr2 = dict(pr); pr2["aperture"]=0.3; pr2["focal"]=8.0; cmp("c1 dof", sc, pr2)
sc, pr = scenes.config_c2(480,270,max_depth=8); cmp("c2 480x270 d8", sc, pr)
sc, pr = scenes.config_c3(160,90,max_depth=1, n=2000); cmp("c3 chain 2000", sc, pr)
sc, pr = scenes.config_c5(192,108,max_depth=4, n=20000); cmp("c5 20k d4", sc, pr)
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc); d.resize(1920,1080); d.count_rays(True)
for f in range(3):
    p = dict(pr); p["seed"] = (0.137+0.1*f, 0.731); d.render(p); d.sync(); s=d.stats(); print("headline frame", f, s.kernel_ms_last, "ms rays", s.rays, "lds", s.lds_bytes, "stack", s.stack_entries, flush=True)
