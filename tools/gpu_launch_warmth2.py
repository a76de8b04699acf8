"""Diagnostic: does a caller-owned (torch) stream / accumulator change what a launch of 20 frames costs?"""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'opengl-raytracer_amd/python')
import torch
torch.cuda.init()
from glrt_amd import scenes, device, host
sc, pr = scenes.config_headline()
W, H = pr["width"], pr["height"]
seeds = lambda f0, k: [host.frame_seed(f0 + i) for i in range(k)]
def make(mode):
    d = device.Device(); d.upload_scene(sc); d.resize(W, H); d.count_rays(False)
    keep = []
    if mode >= 1:
        acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"); d.bind_accum(acc.data_ptr(), W * 16, H); keep.append(acc)
    if mode >= 2:
        st = torch.cuda.Stream(); torch.cuda.set_stream(st); d.set_stream(st.cuda_stream); keep.append(st)
    return d, keep
for mode, name in ((0, "context's own stream and accumulator"), (1, "torch accumulator bound"), (2, "torch accumulator and torch stream")):
    d, keep = make(mode)
    def launch(k, f0=0):
        d.render_frames(pr, seeds(f0, k)); d.sync(); return d.stats().kernel_ms_last
    launch(20); launch(20)
    a = [launch(20, 100 * i) for i in range(4)]
    b = []
    for rep in range(3):
        launch(20, 3); launch(5, 11); torch.cuda.synchronize(); torch.cuda.synchronize()
        b.append(launch(20, 2000 + 50 * rep))
    c = []
    for rep in range(3):
        launch(20, 3); launch(20, 11); launch(5, 5); d.reset_stats(); torch.cuda.synchronize()
        c.append(launch(20, 2000 + 50 * rep))
    print(f"{name:42s} steady " + " ".join(f"{x:.2f}" for x in a) + " | 20,5,sync,20: " + " ".join(f"{x:.2f}" for x in b) + " | 20,20,5,reset_stats,sync,20: " + " ".join(f"{x:.2f}" for x in c))
    if mode >= 2: d.set_stream(0)
    if mode >= 1: d.bind_accum(0, 0, 0)
    d.close()
