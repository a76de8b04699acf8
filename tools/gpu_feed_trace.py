"""What a burst of 96 glrtx_render calls is on the device: run under `rocprofv3 --kernel-trace --stats -- python3 tools/gpu_feed_trace.py [nofeed]` and read the kernel
statistics -- with fed launches the render kernel is dispatched twice (the first frame alone, then one launch that takes the other 95), without them 96 times."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'opengl-raytracer_amd/python')
if len(sys.argv) > 1 and sys.argv[1] == "nofeed":
    os.environ["GLRTX_NO_FEED"] = "1"
from glrt_amd import scenes, device, host
sc, pr = scenes.config_headline()
d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
ps = [device.make_params(dict(pr, seed=host.frame_seed(f))) for f in range(96)]
for p in ps:
    d.render(p)
d.sync()
st = d.stats()
print(f"96 calls: kernel launches {st.kernel_launches}, frames appended to a running launch {st.feed_appended}, render kernel time {st.kernel_ms_total:.2f} ms")
