import sys, time, faulthandler; faulthandler.enable(); sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device
print("build scene", flush=True)
sc, pr = scenes.config_c1(64,64,max_depth=1)
print("scene ok", flush=True)
from oracle import pt_oracle
acc_o, rays_o = pt_oracle.render(sc, pr); print("oracle ok", rays_o, flush=True)
d = device.Device(); print("create ok", flush=True)
d.upload_scene(sc); print("upload ok", flush=True)
d.resize(64,64); print("resize ok", flush=True)
d.count_rays(True)
d.render(pr); print("render ok", flush=True)
d.sync(); print("sync ok", flush=True)
s = d.stats(); print("stats ok", s.rays, s.kernel_ms_last, s.lds_bytes, s.stack_entries, flush=True)
a = d.read_accum(); print("read ok", a.mean(), acc_o.mean(), flush=True)
d.close(); print("close ok", flush=True)
