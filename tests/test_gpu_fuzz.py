"""Randomised and adversarial scenes: the HIP path (all three kernel forms where cheap, one launch per frame and frames in
flight) against the oracle, bit for bit, ray counts included.  Seeds are fixed: failures are reproducible."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from glrt_amd import host, scenes

pytestmark = pytest.mark.gpu


from fuzz_scenes import CASES, case_scene_and_params, fuzz_scene as _scene


@pytest.mark.parametrize("pair_fetch", ["0", "1", "2"], ids=["lane-fetch", "pair-fetch", "alternating-fetch"])  # every form of the wavefront kernel's node fetch (GLRTX_PAIR_FETCH)
@pytest.mark.parametrize("case", CASES, ids=[f"seed{c[0]}" for c in CASES])
def test_fuzz_scene_matches_oracle(gpu_device, monkeypatch, case, pair_fetch):
    from oracle import pt_oracle
    monkeypatch.setenv("GLRTX_PAIR_FETCH", pair_fetch)
    seed, w, h = case[0], case[3], case[4]
    scene, params = case_scene_and_params(case)
    seeds = [host.frame_seed(100 + f) for f in range(3)]
    ref, ref_rays = None, 0
    for sd in seeds:
        ref, n = pt_oracle.render(scene, dict(params, seed=sd), accum=ref)
        ref_rays += n
    d = gpu_device
    d.upload_scene(scene); d.set_partition(0, 1, 16); d.resize(w, h)
    try:
        for variant in (2, 0, 1):
            d.set_variant(variant); d.clear(); d.reset_stats(); d.count_rays(True)
            for sd in seeds:
                d.render(dict(params, seed=sd))
            d.sync()
            assert_bit_equal(d.read_accum(), ref, f"seed {seed} variant {variant}")
            assert d.stats().rays == ref_rays
        d.set_variant(2); d.clear(); d.reset_stats()
        d.render_frames(params, seeds); d.sync()
        assert_bit_equal(d.read_accum(), ref, f"seed {seed} frames in flight")
        assert d.stats().rays == ref_rays
        for variant in (2, 1):  # the compilations WITHOUT ray counting (variant 2's is what bench.py times)
            d.set_variant(variant); d.clear(); d.reset_stats(); d.count_rays(False)
            d.render_frames(params, seeds); d.sync()
            assert_bit_equal(d.read_accum(), ref, f"seed {seed} variant {variant}, kernel without ray counting")
            assert d.stats().rays == 0
    finally:
        d.set_variant(2); d.count_rays(True)


import os


@pytest.mark.parametrize("seed", range(1000, 1000 + int(os.environ.get("GLRT_FUZZ_SEEDS", "24"))))
def test_fuzz_random_parameters(gpu_device, monkeypatch, seed):
    """Parameters drawn from the seed: triangle count, tree builder, image size, depth, samples, lens, flags -- and the form of the node fetch."""
    from oracle import pt_oracle
    rng = np.random.default_rng(seed)
    monkeypatch.setenv("GLRTX_PAIR_FETCH", "012"[seed % 3])
    flags = {k: bool(rng.integers(0, 4) == 0) for k in ("duplicates", "degenerate", "axis_aligned")}
    n_tri = int(rng.integers(1, 400))
    bvh = ("sah", "lbvh", "chain")[int(rng.integers(0, 3))] if n_tri < 120 else ("sah", "lbvh")[int(rng.integers(0, 2))]
    w, h = int(rng.integers(1, 70)), int(rng.integers(1, 50))
    depth, spp = int(rng.integers(0, 17)), int(rng.integers(1, 4))
    aperture = float(rng.choice([0.0, 0.0, 0.1]))
    scene = _scene(seed, n_tri, bvh, **flags)
    eye = tuple(float(v) for v in rng.uniform(-3.5, 3.5, 3))
    if flags["axis_aligned"]:
        eye = (0.0, 0.0, 3.0)
    c2w, s2c = scenes.camera(eye, (0, 0, 0), (0, 1, 0), float(rng.uniform(20, 90)), w, h, 0.1, 100.0)
    params = scenes.make_params(c2w, s2c, w, h, depth, spp, aperture=aperture, focal=3.0)
    seeds = [host.frame_seed(int(rng.integers(0, 10_000)) + f) for f in range(2)]
    ref, ref_rays = None, 0
    for sd in seeds:
        ref, n = pt_oracle.render(scene, dict(params, seed=sd), accum=ref)
        ref_rays += n
    d = gpu_device
    d.upload_scene(scene); d.set_partition(0, 1, 16); d.resize(w, h)
    d.clear(); d.reset_stats(); d.count_rays(True)
    d.render_frames(params, seeds); d.sync()
    assert_bit_equal(d.read_accum(), ref, f"seed {seed}: {n_tri} tris {bvh} {w}x{h} depth {depth} spp {spp} {flags}")
    assert d.stats().rays == ref_rays
    d.clear(); d.reset_stats(); d.count_rays(False)  # the compilation bench.py times
    d.render_frames(params, seeds); d.sync()
    assert_bit_equal(d.read_accum(), ref, f"seed {seed}, kernel without ray counting")
    d.count_rays(True)
