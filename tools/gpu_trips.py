"""Diagnostic (GPU box, lib built with -DGLRTX_PHASE_STATS): per-trip log of sampled workgroups of pt_render_wgwf."""
import sys, ctypes as C; sys.path.insert(0,'.'); sys.path.insert(0,'opengl-raytracer_amd/python')
import numpy as np
from glrt_amd import scenes, device, host
device.lib_path = lambda: device.LIB_DIR / "libglrtx_phase.so"
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc); d.resize(1920, 1080)
L = device.lib(); out = (C.c_uint * (16 * 64 * 4))()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # frames per launch
d.render_frames(pr, [host.frame_seed(f) for f in range(B)]); d.sync(); L.glrtx_debug_trip_log(out)
d.render_frames(pr, [host.frame_seed(B + f) for f in range(B)]); d.sync(); L.glrtx_debug_trip_log(out)
print("frames per launch", B, "kernel ms", d.stats().kernel_ms_last)
lg = np.array(list(out), np.int64).reshape(16, 64, 4)
st = lg[:, 0, 1] * 16; en = lg[:, 0, 2] * 16; base = st.min()
for g in range(16): print(f"workgroup {g*64:4d}: start +{(st[g]-base)/2400:7.1f} us  end +{(en[g]-base)/2400:7.1f} us  trips {lg[g,0,0]}")
for g in (0, 5, 11):
    n = lg[g, 0, 0]
    print(f"workgroup {g*64}: {n} trips, sum traverse {lg[g,1:n+1,2].sum()/2400:.0f} us, sum shade {lg[g,1:n+1,3].sum()/2400:.0f} us (at 2.4 GHz; s_memtime may tick at 100 MHz -> see ratio)")
    for t in range(1, n + 1):
        r, p, ct, cs = lg[g, t]
        print(f"  trip {t:2d}: rays {r:5d} paths {p:5d}  traverse {ct:8d} cyc ({ct/max(r,1):7.1f}/ray)  shade {cs:7d} cyc")
print("first trips of the launch, all sampled workgroups: traverse cycles per ray (rays) | shade cycles per path")
for g in range(16):
    n = lg[g, 0, 0]
    print(f"  wg {g*64:4d}: " + "  ".join(f"{lg[g,t,2]/max(lg[g,t,0],1):6.0f} ({lg[g,t,0]:4d}) | {lg[g,t,3]/max(lg[g,t,1],1):4.0f}" for t in range(1, min(n, 8) + 1)))
tot = lg[:, 1:, :].reshape(-1, 4)
tot = tot[tot[:, 1] > 0]
print("all sampled: trips", len(tot), "traverse cycles", tot[:,2].sum(), "shade", tot[:,3].sum())
