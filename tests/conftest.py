"""Shared test plumbing.  `-m "not gpu"` runs here (no GPU); `-m gpu` runs on an MI355X box."""
import pathlib
import subprocess
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
PKG = ROOT / "opengl-raytracer_amd"
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(PKG / "python"))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """Build the CPU-side libraries if missing (the GPU box receives prebuilt .so files)."""
    if not (PKG / "lib" / "libglrt_host.so").exists():
        subprocess.run(["make", "-C", str(PKG), "host"], check=True, capture_output=True)
    if not (ROOT / "oracle" / "_ref" / "libpt_oracle.so").exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "_ref/libpt_oracle.so"], check=True, capture_output=True)


_ensure_built()


def golden_names():
    return sorted(p.stem for p in GOLDEN.glob("*.npz") if not p.stem.startswith(("math_", "screen_")))


def screen_cases():
    """Resolve-pass fixtures (tests/golden/make_golden_screen.py): (name, accum (H, W, 4) float32, gamma, expected bytes (H, W, 4))."""
    out = []
    z = np.load(GOLDEN / "screen_sweep.npz")
    n = 1024
    x = np.linspace(0, 1, n * n * 3, dtype=np.float64).astype(np.float32).reshape(n, n, 3)
    out.append(("sweep_count1", np.concatenate([x, np.ones((n, n, 1), np.float32)], -1), 2.2, z["out1"]))
    out.append(("sweep_count3", np.concatenate([(x * np.float32(3.0)).astype(np.float32), np.full((n, n, 1), 3.0, np.float32)], -1), 2.2, z["out3"]))
    for name in ("screen_random", "screen_render"):
        z = np.load(GOLDEN / f"{name}.npz")
        acc = np.concatenate([z["rgb"], z["count"][..., None]], -1).astype(np.float32)
        for gm, o in zip(z["gammas"], z["out"]):
            out.append((f"{name[7:]}_gamma{float(gm):.1f}", acc, float(gm), o))
    z = np.load(GOLDEN / "screen_extreme.npz")  # round 4 (make_golden_screen_r04.py): finite but hostile texels, gammas 2.2 .. 1e-45 / 3e38: denormals are flushed
    acc = np.concatenate([z["rgb"], z["count"][..., None]], -1).astype(np.float32)
    for i, (gm, o) in enumerate(zip(z["gammas"], z["out"])):
        out.append((f"extreme_{i}_gamma{float(gm):g}", acc, float(gm), o))
    return out


def load_golden(name):
    z = np.load(GOLDEN / f"{name}.npz")
    scene = {k: z[k] for k in ("vert", "tri", "mat", "light", "bvh")}
    w, h, depth, spp = (int(v) for v in z["scalars"])
    sx, sy, ap, fo = (float(v) for v in z["fparams"])
    params = dict(c2w=z["c2w"], s2c=z["s2c"], width=w, height=h, max_depth=depth, n_samples=spp, seed=(sx, sy),
                  aperture=ap, focal=fo)
    rows = tuple(int(v) for v in z["rows"])
    frames = [tuple(float(c) for c in f) for f in z["frames"]]
    return scene, params, rows, frames, z["out_rgb"], z["out_count"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        idx = np.argwhere(~same)
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        raise AssertionError(f"{what}: {len(idx)} of {a.size} values differ bitwise; first at {tuple(idx[0])}: "
                             f"{a[tuple(idx[0])]!r} vs {b[tuple(idx[0])]!r}; max abs diff {np.nanmax(d):.3e}")


@pytest.fixture(scope="session")
def gpu_device():
    from glrt_amd import device
    # torch brings its own copy of the HIP runtime; initialise it BEFORE libglrtx's (the other order leaves torch without a device:
    # "No HIP GPUs are available" in the tests that hand a torch tensor or stream to the C ABI)
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    d = device.Device()
    yield d
    d.close()
