#!/usr/bin/env python3
"""Generate tests/golden/screen_*.npz: the reference's UNMODIFIED resolve pass (src/shaders/screen.vert / screen.frag, read as text
at run time by oracle/glref.py and never written here) on Mesa llvmpipe, drawn into an RGBA8 colour buffer and read back with
glReadPixels(GL_RGBA, GL_UNSIGNED_BYTE) -- what Window::render's second half and saveCurrentFrame do (window.cpp:297-317, :383-388).

Run in the build container only:  make -C oracle && python tests/golden/make_golden_screen.py

A fixture is data: the accumulator contents (RGB32F colour + R32F count; given by a formula in `recipe` where they are a plain sweep),
u_gamma, and the bytes read back (row 0 = bottom row, before saveCurrentFrame's vertical flip).
  screen_sweep      1024x1024, every byte boundary of pow(x, 1/2.2) many times over: x = linspace(0, 1) and the same through count = 3
  screen_random     64x128: random radiances and counts, gamma 2.2 / 1.0 / 2.4 / 0.7, NaN, negative, zero-count and denormal texels
  screen_render     64x64: an actual render (BASELINE config 1, 16 spp, depth 4)
  screen_npot_50x38 non-power-of-two size: the reference samples its accumulators through GL_LINEAR samplers, exact only at power-of-two
                    sizes; white noise is the worst case for that neighbour leak, the fixture records the reference's bytes
"""
from __future__ import annotations

import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "opengl-raytracer_amd" / "python"))

from glrt_amd import scenes  # noqa: E402
from oracle import pt_oracle  # noqa: E402
from oracle.glref import GLRef  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent
g = GLRef()


def sweep_inputs(n=1024):
    x = np.linspace(0, 1, n * n * 3, dtype=np.float64).astype(np.float32).reshape(n, n, 3)
    return x


def main():
    # 1. sweep (inputs by recipe, outputs compress to a few KB because they are monotone)
    x = sweep_inputs()
    one, three = np.ones(x.shape[:2], np.float32), np.full(x.shape[:2], 3.0, np.float32)
    np.savez_compressed(OUT / "screen_sweep.npz", recipe=np.array("rgb = float32(linspace(0, 1, 1024*1024*3, float64)).reshape(1024,1024,3); "
                        "out1: count = 1; out3: rgb*3 (float32 product), count = 3; gamma 2.2"),
                        out1=g.render_screen(x, one, 2.2), out3=g.render_screen((x * np.float32(3.0)).astype(np.float32), three, 2.2),
                        renderer=np.array(g.info()))
    # 2. random + special values
    rng = np.random.default_rng(20261004)
    h, w = 64, 128
    rgb = (rng.random((h, w, 3), dtype=np.float32) ** 3 * 40).astype(np.float32)
    cnt = rng.integers(1, 33, (h, w)).astype(np.float32)
    rgb[0, 0] = [np.nan, -1.0, 0.5]
    cnt[1, 1] = 0.0; rgb[1, 1] = [0.0, 1.0, -1.0]        # 0/0 = NaN, 1/0 = inf (interior texel: the sampler leaves it alone), -1/0 = -inf
    rgb[2, 2] = [1e-40, 1e-30, 1e-20]
    rgb[3, 3] = [16.0, 8.0, 4.0]; cnt[3, 3] = 16.0         # exactly 1, 0.5, 0.25
    gammas = np.array([2.2, 1.0, 2.4, 0.7], np.float32)
    outs = np.stack([g.render_screen(rgb, cnt, float(gm)) for gm in gammas])
    np.savez_compressed(OUT / "screen_random.npz", rgb=rgb, count=cnt, gammas=gammas, out=outs, renderer=np.array(g.info()))
    # 3. an actual render
    sc, pr = scenes.config_c1(64, 64, max_depth=4, n_samples=16, subdiv=1)
    acc, _ = pt_oracle.render(sc, pr, threads=4)
    np.savez_compressed(OUT / "screen_render.npz", rgb=acc[..., :3].copy(), count=acc[..., 3].copy(), gammas=np.array([2.2], np.float32),
                        out=g.render_screen(acc[..., :3].copy(), acc[..., 3].copy(), 2.2)[None], renderer=np.array(g.info()))
    # 4. NPOT
    h, w = 38, 50
    rgb = (rng.random((h, w, 3), dtype=np.float32) ** 3 * 40).astype(np.float32)
    cnt = np.full((h, w), 16.0, np.float32)
    ref = g.render_screen(rgb, cnt, 2.2)
    mine = pt_oracle.resolve(np.concatenate([rgb, cnt[..., None]], -1), 2.2)
    np.savez_compressed(OUT / "screen_npot_50x38.npz", rgb=rgb, count=cnt, gammas=np.array([2.2], np.float32), out=ref[None],
                        exact_texel_mismatching_bytes=np.array(int((mine != ref).sum())), renderer=np.array(g.info()))
    for f in sorted(OUT.glob("screen_*.npz")):
        print(f.name, f.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
