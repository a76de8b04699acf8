"""Oracle vs the reference's own shader run live on llvmpipe (only where /root/reference and Mesa's
swrast_dri.so exist, i.e. the build container; skipped on the GPU box)."""
import pathlib

import numpy as np
import pytest

from conftest import assert_bit_equal
from fuzz_scenes import CASES as FUZZ_CASES, case_scene_and_params
from glrt_amd import scenes
from oracle import glref, pt_oracle

pytestmark = pytest.mark.skipif(not glref.reference_available(), reason="reference checkout / Mesa llvmpipe not present")


@pytest.fixture(scope="module")
def gl():
    return glref.GLRef()


CASES = [
    ("c1", dict(width=96, height=64, max_depth=6, n_samples=2), {}),
    ("c1", dict(width=64, height=64, max_depth=3, n_samples=1), dict(aperture=0.2, focal=9.0, seed=(0.91, 0.33))),
    ("c2", dict(width=96, height=54, max_depth=8, n_samples=1, subdiv=1), dict(seed=(0.5, 0.25))),
    ("c3", dict(width=64, height=36, max_depth=2, n=800), {}),
    ("c4", dict(width=64, height=36, max_depth=8, n_samples=4, subdiv=1), {}),
    ("c5", dict(width=64, height=36, max_depth=4, n=5000), dict(seed=(0.123, 0.987))),
]


@pytest.mark.parametrize("cfg,kw,over", CASES)
def test_oracle_bit_exact_vs_live_reference(gl, cfg, kw, over):
    sc, pr = scenes.CONFIGS[cfg](**kw)
    pr = dict(pr, **over)
    rgb, cnt = gl.render_reference(sc, pr)
    acc, rays = pt_oracle.render(sc, pr)
    assert rays >= pr["width"] * pr["height"] * pr["n_samples"]
    assert_bit_equal(acc[..., :3], rgb, "rgb")
    assert_bit_equal(acc[..., 3], cnt, "count")


def test_random_seeds_sweep(gl):
    sc, pr = scenes.config_c1(48, 48, max_depth=5, n_samples=1, subdiv=1)
    rng = np.random.default_rng(5)
    for _ in range(6):
        p = dict(pr, seed=tuple(float(np.float32(v)) for v in rng.uniform(0, 1, 2)))
        rgb, cnt = gl.render_reference(sc, p)
        acc, _ = pt_oracle.render(sc, p)
        assert_bit_equal(acc[..., :3], rgb, f"seed {p['seed']}")


@pytest.mark.parametrize("case", FUZZ_CASES, ids=[f"seed{c[0]}" for c in FUZZ_CASES])
def test_oracle_bit_exact_vs_live_reference_on_adversarial_scenes(gl, case):
    """The scenes of tests/test_gpu_fuzz.py (ties, degenerate and axis-aligned geometry, large distances, 1-2 triangles)
    through the reference's own shader: pins the oracle where the GPU tests rely on it."""
    sc, pr = case_scene_and_params(case)
    rgb, cnt = gl.render_reference(sc, pr)
    acc, _ = pt_oracle.render(sc, pr)
    assert_bit_equal(acc[..., :3], rgb, "rgb")
    assert_bit_equal(acc[..., 3], cnt, "count")


def test_resolve_restatement_vs_live_screen_shader(gl):
    """The reference's screen.frag drawn into RGBA8 and read back, live, against the oracle's resolve on fresh random accumulators
    (power-of-two size, several gammas) and on a rendered image."""
    rng = np.random.default_rng(99)
    for h, w in ((128, 256), (32, 32)):
        rgb = (rng.random((h, w, 3), dtype=np.float32) ** 2 * 60).astype(np.float32)
        cnt = rng.integers(1, 65, (h, w)).astype(np.float32)
        for gamma in (2.2, 1.8, 1.0, 3.0):
            ref = gl.render_screen(rgb, cnt, gamma)
            got = pt_oracle.resolve(np.concatenate([rgb, cnt[..., None]], -1), gamma)
            assert np.array_equal(got, ref), f"{h}x{w} gamma {gamma}: {int((got != ref).sum())} bytes differ"
    sc, pr = scenes.config_c2(64, 32, max_depth=4, n_samples=8, subdiv=1)
    acc, _ = pt_oracle.render(sc, pr)
    assert np.array_equal(pt_oracle.resolve(acc, 2.2), gl.render_screen(acc[..., :3].copy(), acc[..., 3].copy()))


def test_resolve_restatement_vs_live_screen_shader_on_hostile_accumulators(gl):
    """Finite but hostile texels (denormals, negatives, 1e-30 .. 3e38, counts of 0 / 3e38 / fractions, values scaled by 2^-140 .. 2^119) and gammas from 1e-45
    to 3e38: llvmpipe runs the pass with denormals flushed, and so does the restatement since round 4 (counts and gammas of 3e38 showed the difference).
    Not generated: NaN / infinite texels and negative zeros -- the GL_LINEAR samplers add the neighbouring texels with weight 0, outside the restatement's domain."""
    import sys
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent / "golden"))
    from make_golden_screen_r04 import extreme_accumulator
    acc = extreme_accumulator()
    rng = np.random.default_rng(404)
    for it in range(12):
        a = acc.copy()
        rng.shuffle(a.reshape(-1, 4))  # another arrangement of the same texels
        gamma = float(rng.choice([2.2, 1.0, 0.45, 1e-20, 1e20, 3.0, 1e-45, 3e38, float(rng.uniform(0.1, 5))]))
        ref = gl.render_screen(a[..., :3].copy(), a[..., 3].copy(), gamma)
        got = pt_oracle.resolve(a, gamma)
        assert np.array_equal(got, ref), f"arrangement {it}, gamma {gamma}: {int((got != ref).sum())} bytes differ"


def test_oracle_bit_exact_vs_live_reference_on_hostile_inputs(gl):
    """The inputs that round 4's device fuzzes throw at the kernels -- determinants beyond 2^126 (the reciprocal is a denormal: llvmpipe flushes it, no hit), scenes
    scaled by 1e-12 .. 1e12, seeds up to 1e20 / inf / NaN, conductors and albedos and emitters of 0 .. 1e30 / inf / NaN, vertices at infinity and NaN -- through the
    LIVE reference: the oracle is the reference there too, so the device-against-oracle tests of the same inputs (tests/test_gpu_parity.py) mean what they say."""
    from hostile_cases import cases
    n = 0
    for tag, sc, p in cases():
        rgb, cnt = gl.render_reference(sc, p)
        acc, _ = pt_oracle.render(sc, p)
        a = np.concatenate([rgb, cnt[..., None]], -1)
        same = (a.view(np.uint32) == acc.view(np.uint32)) | (np.isnan(a) & np.isnan(acc))
        assert same.all(), f"{tag}: {int((~same).any(-1).sum())} pixels differ"
        n += 1
    assert n >= 50


@pytest.mark.parametrize("seed", range(32))
def test_oracle_bit_exact_vs_live_reference_on_shadow_hostile_scenes(gl, seed):
    """Lights flush in the faces of their ancestors' boxes, sliver light triangles, receivers that run up to the lights' edges (grazing light samples), distances
    1 .. 100, the scene translated to |coordinates| up to 1e7 (tests/fuzz_scenes.py: shadow_hostile) -- what the shadow rays' search is most sensitive to -- through
    the LIVE reference: the oracle is the reference there, so the device tests of the same scenes (tests/test_gpu_fuzz.py) mean what they say."""
    from fuzz_scenes import shadow_hostile
    tag, sc, pr = shadow_hostile(seed)
    rgb, cnt = gl.render_reference(sc, pr)
    acc, _ = pt_oracle.render(sc, pr)
    assert_bit_equal(acc[..., :3], rgb, tag)
    assert_bit_equal(acc[..., 3], cnt, tag)
