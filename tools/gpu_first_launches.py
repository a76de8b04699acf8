"""Diagnostic: what in bench.py's untimed preamble makes its timed launch of 20 frames slower than the steady state of the same launch?"""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'opengl-raytracer_amd/python')
import torch
torch.cuda.init()
from glrt_amd import scenes, device, host
sc, pr = scenes.config_headline()
W, H = pr["width"], pr["height"]
seeds = lambda f0, k: [host.frame_seed(f0 + i) for i in range(k)]
def fresh(torch_side):
    d = device.Device(); d.upload_scene(sc); d.resize(W, H); d.count_rays(False)
    keep = []
    if torch_side:
        acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"); d.bind_accum(acc.data_ptr(), W * 16, H)
        st = torch.cuda.Stream(); torch.cuda.set_stream(st); d.set_stream(st.cuda_stream); keep = [acc, st]
    return d, keep
def L(d, k, f0): d.render_frames(pr, seeds(f0, k))
def timed(d):
    L(d, 20, 5); d.sync(); return d.stats().kernel_ms_last
for name, torch_side, counting, clones in (("plain context, no counting pass", False, False, False), ("plain context, counting pass first", False, True, False),
                                           ("torch stream + accumulator, counting pass", True, True, False), ("... and clone / compare / zero between", True, True, True)):
    res, steady = [], []
    for rep in range(2):
        d, keep = fresh(torch_side)
        if counting: d.count_rays(True)
        L(d, 20, 5)
        if counting: d.count_rays(False)
        if clones:
            c = keep[0].clone(); keep[0].zero_()
        L(d, 20, 5)
        if clones:
            same = (keep[0].view(torch.int32) == c.view(torch.int32)).all(); del c; keep[0].zero_()
        L(d, 5, 0); d.sync(); torch.cuda.synchronize()
        res.append(timed(d))
        steady.append(min(timed(d) for _ in range(4)))
        if torch_side: d.set_stream(0); d.bind_accum(0, 0, 0)
        d.close()
    print(f"{name:52s} timed launch " + " ".join(f"{x:.2f}" for x in res) + "   steady state afterwards " + " ".join(f"{x:.2f}" for x in steady))
