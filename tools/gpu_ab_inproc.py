"""A/B timing of several builds of libglrtx.so INSIDE ONE PROCESS: every variant gets a context of its own on the same device, and the timed launches go round the
variants in turn (A B C A B C ...), a launch behind the other's end.  tools/gpu_abx.py runs one process per variant and pass, and on this pool processes (and boxes)
differ by up to 3 % for one binary (profiles/r04_ab_hit_park.txt); here both variants see the same box, clock and moment, and differences of a few tenths of a percent
show.  Each variant is a copy of glrt_amd.device loaded as a module of its own (ctypes loads each library RTLD_LOCAL).

    python tools/gpu_ab_inproc.py [--config headline|c2..c5] [--frames 20] [--rounds 40] NAME=path/to/lib.so[,ENV=VAL...] ...

ENV=VAL pairs are set while that variant's library is loaded and its context created (GLRTX_* switches are read at launch: set while that variant launches).
Prints per variant: median / mean ms per frame, and the median of the per-round differences to the first variant with its spread."""
import hashlib
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
import pathlib  # noqa: E402

import glrt_amd  # noqa: E402,F401
from glrt_amd import host, scenes  # noqa: E402


def load_device_module(tag, lib):
    spec = importlib.util.spec_from_file_location(f"glrt_amd.device_{tag}", os.path.join(ROOT, "opengl-raytracer_amd", "python", "glrt_amd", "device.py"),
                                                  submodule_search_locations=None)
    m = importlib.util.module_from_spec(spec)
    m.__package__ = "glrt_amd"
    sys.modules[spec.name] = m
    spec.loader.exec_module(m)
    m.lib_path = lambda: pathlib.Path(lib)
    return m


class EnvSet:
    def __init__(self, kv): self.kv, self.old = kv, {}
    def __enter__(self):
        for k, v in self.kv.items(): self.old[k] = os.environ.get(k); os.environ[k] = v
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def main():
    a = sys.argv[1:]
    config, frames, rounds, variants = "headline", 20, 40, []
    one_stream = False
    while a:
        x = a.pop(0)
        if x == "--config": config = a.pop(0)
        elif x == "--frames": frames = int(a.pop(0))
        elif x == "--rounds": rounds = int(a.pop(0))
        elif x == "--one-stream": one_stream = True  # every context renders on ONE stream created here (default: each on its context's own)
        else: variants.append(x)
    sc, pr = scenes.CONFIGS[config]()
    V = []
    common = None
    if one_stream:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        common = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(common), 1) == 0  # hipStreamNonBlocking
    for i, v in enumerate(variants):
        name, rest = v.split("=", 1)
        parts = rest.split(",")
        env = dict(kv.split("=", 1) for kv in parts[1:])
        with EnvSet(env):
            m = load_device_module(f"{i}", os.path.join(ROOT, parts[0]))
            d = m.Device(); d.upload_scene(sc); d.resize(pr["width"], pr["height"])
            if common is not None: d.set_stream(common.value)
            d.count_rays(True); d.reset_stats()
            d.render_frames(pr, [host.frame_seed(i_) for i_ in range(frames)]); d.sync()
            st = d.stats()
            h = hashlib.sha1(np.ascontiguousarray(d.read_accum()).view(np.uint8)).hexdigest()
            d.count_rays(False)
        V.append(dict(name=name, env=env, d=d, rays=int(st.rays), sha=h, ms=[]))
    same = all(v["sha"] == V[0]["sha"] and v["rays"] == V[0]["rays"] for v in V)
    print(f"{len(V)} variants, {config}, {frames} frames per launch, {rounds} rounds; images and ray counts {'identical' if same else 'DIFFER'}", flush=True)
    for r in range(rounds + 2):
        order = list(range(len(V)))
        order = order[r % len(V):] + order[:r % len(V)]  # rotate who goes first
        for k in order:
            v = V[k]
            with EnvSet(v["env"]):
                v["d"].render_frames(pr, [host.frame_seed(frames * (r + 1) + i_) for i_ in range(frames)]); v["d"].sync()
            if r >= 2: v["ms"].append(v["d"].stats().kernel_ms_last / frames)
    base = np.asarray(V[0]["ms"])
    for v in V:
        ms = np.asarray(v["ms"]); diff = (ms - base) / base * 100.0
        q = np.percentile(diff, [25, 50, 75])
        print(f"{v['name']:24s} median {np.median(ms):.4f}  mean {ms.mean():.4f} ms/frame   vs {V[0]['name']}: median of per-round differences {q[1]:+.2f} %  (quartiles {q[0]:+.2f} .. {q[2]:+.2f})", flush=True)


if __name__ == "__main__":
    main()
