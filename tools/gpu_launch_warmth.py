"""Diagnostic (GPU box): what a launch of n frames costs depending on what the device did just before it -- the driver times ONE launch of 20 frames behind a warm-up launch of 5.
Usage: gpu_launch_warmth.py [frames]"""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'opengl-raytracer_amd/python')
from glrt_amd import scenes, device, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"]); d.count_rays(False)
seeds = lambda f0, k: [host.frame_seed(f0 + i) for i in range(k)]

def launch(k, f0=0):
    d.render_frames(pr, seeds(f0, k)); d.sync(); return d.stats().kernel_ms_last

launch(n); launch(n)  # buffers exist
print(f"steady state, {n} frames per launch, back to back:", " ".join(f"{launch(n, 100 * i):.2f}" for i in range(5)), "ms")
for warm, pause in ((5, 0.0), (5, 0.05), (20, 0.0), (48, 0.0), (0, 0.5), (0, 2.0), (5, 0.0), (1, 0.0)):
    res = []
    for rep in range(3):
        time.sleep(1.0)            # the device has been idle
        if warm: launch(warm, 7)
        if pause: time.sleep(pause)
        res.append(launch(n, 1000 + 50 * rep))
    print(f"idle 1 s, warm-up launch of {warm:2d} frames, pause {pause * 1e3:4.0f} ms, then {n} frames: " + " ".join(f"{x:.2f}" for x in res) + " ms")

print("no idle time anywhere (launches back to back, the last one timed):")
for seq in ((20, 20), (5, 20), (20, 5, 20), (5, 5, 5, 5, 20), (48, 5, 20), (20, 10, 20), (20, 1, 20), (20, 19, 20), (20, 21, 20)):
    res = []
    for rep in range(3):
        launch(20, 3)
        for k in seq[:-1]:
            t = launch(k, 11)
        res.append(launch(seq[-1], 2000 + 50 * rep))
    print(f"  launches of 20 | " + " | ".join(str(k) for k in seq) + " frames: the last takes " + " ".join(f"{x:.2f}" for x in res) + " ms")

print("a pause between two launches of 20 frames (device warm before):")
for pause_ms in (0.0, 0.1, 0.3, 1.0, 3.0, 10.0, 30.0):
    res = []
    for rep in range(3):
        launch(20, 3); launch(20, 5)
        if pause_ms:
            t_end = time.perf_counter() + pause_ms * 1e-3
            while time.perf_counter() < t_end: pass
        res.append(launch(20, 3000 + 50 * rep))
    print(f"  pause {pause_ms:5.1f} ms: " + " ".join(f"{x:.2f}" for x in res) + " ms")
