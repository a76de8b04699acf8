"""Round 4: one more resolve-pass fixture from the reference's screen.{vert,frag} on llvmpipe (run in the build container; same runner as make_golden_screen.py).

screen_extreme.npz: a 64 x 64 accumulator of finite but hostile values -- denormals, 1e-30 .. 3e38, negatives, counts of 0, 3e38 and fractions, values scaled by
2^-140 .. 2^119 -- resolved with gammas 2.2, 0.45, 1e-20, 1e20, 3e38 and 1e-45.  What it pins: the pass runs with denormals flushed like every llvmpipe fragment
shader (a quotient rgb / count or an exponent 1 / gamma below FLT_MIN is 0: pow(x, 0) = 1), which neither the oracle nor the device did for this pass before
round 4.  Not in it, because the reference's GL_LINEAR samplers add the neighbouring texels with weight 0: NaN / infinite texels (0 * inf poisons the pixels
around them) and negative zeros (a count of -0 comes out as +0)."""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import glref  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent


def extreme_accumulator():
    rng = np.random.default_rng(20261004 + 4)
    h = w = 64
    specials = np.array([0.0, 1e-45, 1e-40, 1.1754944e-38, 1e-30, 1e-6, 0.5, 1.0, 1.0000001, 2.0, 100.0, 1e30, 3e38, -1.0, -1e-30], np.float32)
    acc = rng.uniform(0, 4, (h, w, 4)).astype(np.float32)
    acc[..., 3] = rng.integers(0, 5, (h, w)).astype(np.float32)
    m = rng.uniform(0, 1, acc.shape) < 0.15
    acc[m] = specials[rng.integers(0, specials.size, int(m.sum()))]
    e = rng.uniform(0, 1, acc.shape) < 0.2
    with np.errstate(all="ignore"):
        acc[e] = (acc[e] * np.float32(2.0) ** rng.integers(-140, 120, int(e.sum())).astype(np.float32)).astype(np.float32)
    acc[~np.isfinite(acc)] = 1.0
    acc[(acc == 0) & np.signbit(acc)] = 0.0               # no negative zeros ...
    acc[(np.abs(acc) < 1.1754944e-38) & (acc < 0)] = 0.0   # ... and no negative denormals, which flush to one
    return acc


def main():
    g = glref.GLRef()
    acc = extreme_accumulator()
    gammas = np.array([2.2, 0.45, 1e-20, 1e20, 3e38, 1e-45], np.float32)
    outs = np.stack([g.render_screen(acc[..., :3].copy(), acc[..., 3].copy(), float(gm)) for gm in gammas])
    np.savez_compressed(OUT / "screen_extreme.npz", rgb=acc[..., :3].copy(), count=acc[..., 3].copy(), gammas=gammas, out=outs, renderer=np.array(g.info()))
    print("screen_extreme.npz", (OUT / "screen_extreme.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
