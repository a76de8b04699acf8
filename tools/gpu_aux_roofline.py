"""The two HBM-bound kernels of the path on the GPU box: accumulate_planes_kernel (frames in flight: sample planes -> accumulator) and
resolve_kernel (screen.frag:15-25 + RGBA8), at 1920x1080 and 3840x2160, GB/s against the 8 TB/s HBM peak.  Device times come from HIP
events on the launch stream (glrtx_stats.accumulate_ms_total, .resolve_ms_last).  Writes gpurun_out/r06_aux_kernels.json (copied to profiles/)."""
import json, sys; sys.path.insert(0, '.'); sys.path.insert(0, 'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
PEAK = 8000.0
out = {"device": "MI355X", "peak_gb_s": PEAK, "rows": []}
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc)
for (w, h) in ((1920, 1080), (3840, 2160)):
    d.resize(w, h)
    p = dict(pr, width=w, height=h)
    c2w, s2c = scenes.camera((0, 5, 16), (0, 2.0, 0), (0, 1, 0), 40.0, w, h)
    p = dict(scenes.make_params(c2w, s2c, w, h, 8, 1))
    for planes in (8, 16, 24):
        ms = []
        for it in range(4):
            d.reset_stats()
            d.render_frames(p, [host.frame_seed(it * planes + f) for f in range(planes)]); d.sync()
            ms.append(d.stats().accumulate_ms_total)
        t = float(np.median(ms[1:]))
        by = w * h * 16 * (planes + 2)
        out["rows"].append({"kernel": "accumulate_planes_kernel", "size": f"{w}x{h}", "planes": planes, "ms": round(t, 4), "bytes": by,
                            "gb_s": round(by / t / 1e6, 1), "frac_of_hbm_peak": round(by / t / 1e6 / PEAK, 4)})
    ms = []
    for it in range(6):
        d.resolve_rgba8(2.2, True); ms.append(d.stats().resolve_ms_last)
    t = float(np.median(ms[1:]))
    by = w * h * 20
    tb = d.resolve_burst_ms(2.2, 32)
    out["rows"].append({"kernel": "resolve_kernel", "size": f"{w}x{h}", "ms": round(tb, 4), "bytes": by, "gb_s": round(by / tb / 1e6, 1),
                        "frac_of_hbm_peak": round(by / tb / 1e6 / PEAK, 4), "how": "32 launches back to back between one pair of events, per launch",
                        "single_launch_between_two_events_ms": round(t, 4)})
for r in out["rows"]: print(r)
json.dump(out, open("gpurun_out/r06_aux_kernels.json", "w"), indent=1)
