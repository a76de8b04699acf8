"""sin/cos and the RNG recurrence of the restatement against values the GL implementation computed
(diagnostic-shader sweeps, tests/golden/math_*.npz).  Bit-exact."""
import numpy as np

from conftest import GOLDEN, assert_bit_equal
from oracle import pt_oracle


def test_sincos_bit_exact_against_gl_sweep():
    z = np.load(GOLDEN / "math_sincos.npz")
    s, c = pt_oracle.sincos(z["x"])
    assert_bit_equal(s, z["sin"], "sin")
    assert_bit_equal(c, z["cos"], "cos")


def test_sincos_close_to_libm():
    x = np.linspace(-92, 92, 100001).astype(np.float32)
    s, c = pt_oracle.sincos(x)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 3e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 3e-7


def test_rand_recurrence_bit_exact_against_gl():
    z = np.load(GOLDEN / "math_rand.npz")
    st = z["state_seed"]
    out = z["out"]
    # rand_stream seeds its state as ((px + .5) / W, (py + .5) / H): choose W = H = 1 and fractional "pixels"
    # is not possible through that entry, so drive the recurrence through the pixel-like subset only.
    got = 0
    for i in range(0, st.shape[0], 3):
        sx, sy = st[i, 0], st[i, 1]
        px, py = int(np.floor(sx * 1920)), int(np.floor(sy * 1080))
        if np.float32(np.float32(px + 0.5) / np.float32(1920)) != sx or \
           np.float32(np.float32(py + 0.5) / np.float32(1080)) != sy:
            continue
        r = pt_oracle.rand_stream(1920.0, 1080.0, px, py, (float(st[i, 2]), float(st[i, 3])), 3)
        assert_bit_equal(r, out[i, :3], f"rand stream {i}")
        got += 1
    assert got > 1000


def test_rand_in_unit_interval_and_deterministic():
    a = pt_oracle.rand_stream(1920.0, 1080.0, 17, 901, (0.137, 0.731), 4096)
    b = pt_oracle.rand_stream(1920.0, 1080.0, 17, 901, (0.137, 0.731), 4096)
    assert np.array_equal(a, b)
    assert a.min() >= 0.0 and a.max() < 1.0
    assert 0.4 < a.mean() < 0.6
