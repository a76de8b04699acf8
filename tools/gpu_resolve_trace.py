"""The resolve kernel's own duration (dispatch begin to end, rocprofv3 --kernel-trace --stats) at 1080p and 4K: run as
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/gpu_resolve_trace.py
Next to it the per-launch time of a train of 32 launches between one pair of HIP events (which contains the gaps between consecutive dispatches of one stream)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'opengl-raytracer_amd/python')
from glrt_amd import scenes, device, host
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc)
for (w, h) in ((1920, 1080), (3840, 2160)):
    d.resize(w, h)
    c2w, s2c = scenes.camera((0, 5, 16), (0, 2.0, 0), (0, 1, 0), 40.0, w, h)
    p = dict(scenes.make_params(c2w, s2c, w, h, 8, 1))
    d.render_frames(p, [host.frame_seed(f) for f in range(4)]); d.sync()
    print(f"{w}x{h}: {d.resolve_burst_ms(2.2, 32) * 1e3:.2f} us per launch in a train of 32 (HIP events)", flush=True)
