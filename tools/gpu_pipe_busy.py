"""Diagnostic (GPU box): one launch per frame (overlapped single-frame launches) while the caller keeps streams of its own busy with small kernels.
Prints wall ms per frame, the render kernels' own device time, and how many slots were resident together.  Usage: gpu_pipe_busy.py [frames]"""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'opengl-raytracer_amd/python')
import torch
torch.cuda.init()
from glrt_amd import scenes, device, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"]); d.count_rays(False)

def run(n, n_streams, every=1, size=1 << 16):
    d.clear(); d.reset_stats()
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    xs = [torch.zeros(size, device="cuda") for _ in streams]
    d.sync(); torch.cuda.synchronize()
    t = time.perf_counter()
    for f in range(n):
        d.render(dict(pr, seed=host.frame_seed(f)))
        if f % every == 0:
            for s_, x in zip(streams, xs):
                with torch.cuda.stream(s_):
                    x.add_(1.0)
    t_issue = time.perf_counter() - t
    d.sync(); torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / n * 1e3
    st = d.stats()
    return f"wall {wall:.3f} ms/frame (issued in {t_issue / n * 1e3:.3f}), render kernels {st.kernel_ms_total / max(st.kernel_launches, 1):.3f} ms each, accumulate {st.accumulate_ms_total / max(st.kernel_launches, 1):.3f}, resident max {st.pipe_resident_max} of {st.pipe_slots} slots"

run(8, 0)
run(8, 4)  # the caller's kernel is loaded, its streams have launched once (a code-object load stalls the device: not what is measured here)
for label, args in (("quiet", (n, 0)), ("1 caller stream", (n, 1)), ("2 caller streams", (n, 2)), ("2 caller streams, a kernel every 4th frame", (n, 2, 4)),
                    ("4 caller streams", (n, 4)), ("2 caller streams, 16 M-element kernels", (n, 2, 1, 1 << 24)), ("quiet again", (n, 0))):
    print(f"{label:45s} {run(*args)}")
