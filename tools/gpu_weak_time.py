"""What bench.py's N-rank step costs one rank, measured on ONE GPU: rank 0's share (1/N of the rows in interleaved stripes) of
N x B frames per launch, for N = 1, 2, 4, 8 -- the compute side of the weak-scaling curve, without the gather."""
import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc)
for B in (10, 20, 24):
    row = []
    for world in (1, 2, 4, 8):
        d.set_partition(0, world, 16); d.resize(1920, 1080)
        ts = []
        for it in range(4):
            d.render_frames(pr, [host.frame_seed(it * B * world + f) for f in range(B * world)]); d.sync(); ts.append(d.stats().kernel_ms_last)
        row.append(f"N={world}: {np.median(ts[1:]) / B:.3f}")
    print(f"{B} steps per launch, ms per step on rank 0:", "   ".join(row), flush=True)
