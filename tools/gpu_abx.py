"""A/B timing of several builds of libglrtx.so (and/or environment settings) on the headline workload.

    python tools/gpu_abx.py [--config headline|c2..c5|rand:N] [--frames 16] [--rounds 6] [--repeat 1] NAME=path/to/lib.so[,ENV=VAL...] ...

Each variant runs in its own child process (one libglrtx per process), sequentially: an untimed counting launch
(ray count + image checksum), then `rounds` timed launches of `frames` frames in flight.  Prints ms/frame (median,
min) per variant, the ray counts, and whether every variant's image is bit-identical to the first one's.
Child mode: --child NAME LIB."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(lib, config, frames, rounds):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
    import pathlib
    import numpy as np
    from glrt_amd import device, host, scenes
    device.lib_path = lambda: pathlib.Path(lib)
    if config.startswith("rand:"):  # rand:N -- N random triangles at config 5's density and camera distance (tree-size sweeps)
        n = int(config[5:]); ext = 10.0 * (n / 100_000.0) ** (1.0 / 3.0)
        sc, pr = scenes._random_tri_scene(n, 20260102, ext, 3.4 * ext, 1920, 1080, 4, 1, "sah")
    else:
        sc, pr = scenes.CONFIGS[config]()
    if os.environ.get("ABX_PREALLOC_MB"):  # (placement experiments: device memory taken before the library allocates anything)
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        for mb in os.environ["ABX_PREALLOC_MB"].split("+"):
            ptr = ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(int(float(mb) * (1 << 20)))) == 0
    d = device.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
    seeds = lambda f0: [host.frame_seed(f0 + i) for i in range(frames)]
    d.count_rays(True); d.reset_stats()
    d.render_frames(pr, seeds(0)) if frames > 1 else d.render(dict(pr, seed=host.frame_seed(0)))
    d.sync()
    st = d.stats(); rays, untr = int(st.rays), int(st.rays_untraced)
    img = d.read_accum(); h = hashlib.sha1(np.ascontiguousarray(img).view(np.uint8)).hexdigest()
    d.count_rays(False)
    ms = []
    for r in range(rounds + 1):
        d.render_frames(pr, seeds(frames * (r + 1))) if frames > 1 else d.render(dict(pr, seed=host.frame_seed(r + 1)))
        d.sync()
        if r: ms.append(d.stats().kernel_ms_last / frames)
    ms.sort()
    print(json.dumps({"rays": rays, "untraced": untr, "sha1": h, "ms_med": ms[len(ms) // 2], "ms_min": ms[0], "ms_all": [round(x, 4) for x in ms]}), flush=True)


def main():
    a = sys.argv[1:]
    if a and a[0] == "--child":
        return child(a[1], a[2], int(a[3]), int(a[4]))
    config, frames, rounds, repeat, variants = "headline", 16, 6, 1, []
    while a:
        x = a.pop(0)
        if x == "--config": config = a.pop(0)
        elif x == "--frames": frames = int(a.pop(0))
        elif x == "--rounds": rounds = int(a.pop(0))
        elif x == "--repeat": repeat = int(a.pop(0))
        else: variants.append(x)
    # the variants are cycled `repeat` times (A B C A B C ...): clocks drift over a session by more than the differences of interest
    first, res = None, {}
    for rep in range(repeat):
        for v in variants:
            name, rest = v.split("=", 1)
            parts = rest.split(",")
            lib = os.path.join(ROOT, parts[0])
            env = dict(os.environ)
            for kv in parts[1:]:
                k, val = kv.split("=", 1); env[k] = val
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, config, str(frames), str(rounds)], env=env, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                print(f"{name:28s} FAILED rc={r.returncode} {r.stderr[-400:]}", flush=True)
                continue
            o = json.loads(line[-1])
            first = first or o
            same = "same image" if o["sha1"] == first["sha1"] and o["rays"] == first["rays"] else "IMAGE/RAYS DIFFER"
            res.setdefault(name, []).append(o["ms_med"])
            print(f"{name:28s} {o['ms_med']:.4f} ms/frame (min {o['ms_min']:.4f})  rays {o['rays']} untraced {o['untraced']}  {same}  {o['ms_all']}", flush=True)
    if repeat > 1:
        base = None
        for name, v in res.items():
            v = sorted(v); med = v[len(v) // 2]
            base = base or med
            print(f"== {name:25s} median of {len(v)} passes {med:.4f} ms/frame ({(med / base - 1) * 100:+.2f} % vs {list(res)[0]})  passes {[round(x, 4) for x in v]}", flush=True)


if __name__ == "__main__":
    main()
