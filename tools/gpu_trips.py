"""Per-trip log of sampled workgroups of ONE launch (libglrtx_phase.so: g_trip_log -- every 64th workgroup, its first 63 trips... the log keeps the FIRST 63 trips, so the
launch is kept short enough for the drain to be inside them): paths alive, rays, shader clocks of the traverse and shade halves.
    python tools/gpu_trips.py [frames=3]"""
import ctypes as C, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
device.lib_path = lambda: device.LIB_DIR / "libglrtx_phase.so"
os.environ["GLRTX_NO_FEED"] = "1"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc); d.resize(1920, 1080)
L = device.lib(); buf = (C.c_uint * (16 * 64 * 4))()
d.render_frames(pr, [host.frame_seed(f) for f in range(B)]); d.sync(); L.glrtx_debug_trip_log(buf)
d.render_frames(pr, [host.frame_seed(B + f) for f in range(B)]); d.sync(); L.glrtx_debug_trip_log(buf)
print("frames", B, "kernel ms", d.stats().kernel_ms_last)
a = np.frombuffer(buf, np.uint32).reshape(16, 64, 4)
for w in (0, 5, 11):
    n = int(a[w, 0, 0])
    print(f"workgroup {w * 64}: {n} trips logged, start {a[w, 0, 1]} end {a[w, 0, 2]} (clk/16)")
    print("   paths:", " ".join(str(int(x)) for x in a[w, 1:n + 1, 1]))
    print("   rays :", " ".join(str(int(x)) for x in a[w, 1:n + 1, 0]))
    print("   trav kclk:", " ".join(str(int(x) // 1000) for x in a[w, 1:n + 1, 2]))
    print("   shade kclk:", " ".join(str(int(x) // 1000) for x in a[w, 1:n + 1, 3]))
