"""Extension kernel (f4, PARITY UNPINNED) timings on the GPU box: BASELINE's sphere scenes read literally -- "3 spheres + 1 ground plane"
(scenes.config_spheres, analytic spheres staged in LDS) at 256x256 / 1 bounce, 1920x1080 / 4 bounces, 3840x2160 / 8 bounces -- Mrays/s of the
persistent megakernel's extension instantiation, next to the same scene with icospheres on the pinned triangle path (wavefront kernel).
Writes gpurun_out/r03_ext_timing.json."""
import json, sys; sys.path.insert(0, '.'); sys.path.insert(0, 'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
out = {"rows": []}
d = device.Device()
for (w, h, depth) in ((256, 256, 1), (1920, 1080, 4), (3840, 2160, 8)):
    for mode in ("analytic spheres, diffuse/conductor", "analytic spheres, glass + Whitted", "icospheres (subdiv 3) on the pinned triangle path"):
        glass = "glass" in mode
        sub = 3 if "icospheres" in mode else None
        sc, pr, sph = scenes.config_spheres(w, h, max_depth=depth, glass=glass, subdiv=sub)
        flags = (device.EXT_DIELECTRIC | device.EXT_WHITTED) if glass else 0
        d.upload_scene(sc); d.upload_spheres(sph); d.set_extensions(flags); d.set_partition(0, 1, 16); d.resize(w, h)
        d.count_rays(True); d.reset_stats()
        d.render(dict(pr, seed=host.frame_seed(0))); d.sync()
        rays = int(d.stats().rays) - int(d.stats().rays_untraced)
        d.count_rays(False)
        ms = []
        n = 8
        for it in range(4):
            d.reset_stats()
            d.render_frames(pr, [host.frame_seed(1 + it * n + f) for f in range(n)]); d.sync()
            st = d.stats()
            ms.append((st.kernel_ms_total + st.accumulate_ms_total) / n)
        t = float(np.median(ms[1:]))
        row = {"size": f"{w}x{h}", "max_depth": depth, "scene": mode, "kernel": "pt_render_wgwf" if st.variant_last == 2 else "pt_render_persistent<*, EXT>",
               "rays_per_frame": rays, "ms_per_frame": round(t, 4), "mrays_per_s": round(rays / t / 1e3, 1)}
        out["rows"].append(row); print(row, flush=True)
        d.set_extensions(0); d.upload_spheres(None)
json.dump(out, open("gpurun_out/r03_ext_timing.json", "w"), indent=1)
