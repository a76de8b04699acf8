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


@pytest.mark.parametrize("seed", range(int(os.environ.get("GLRT_FUZZ_FIRST", "1000")), int(os.environ.get("GLRT_FUZZ_FIRST", "1000")) + int(os.environ.get("GLRT_FUZZ_SEEDS", "24"))))
def test_fuzz_random_parameters(gpu_device, monkeypatch, seed):
    """Parameters drawn from the seed: triangle count, tree builder, image size, depth, samples, lens, flags -- and the form of the node fetch."""
    from oracle import pt_oracle
    rng = np.random.default_rng(seed)
    monkeypatch.setenv("GLRTX_PAIR_FETCH", "012"[seed % 3])
    flags = {k: bool(rng.integers(0, 4) == 0) for k in ("duplicates", "degenerate", "axis_aligned")}
    n_tri = int(rng.integers(1, 400))
    bvh = ("sah", "lbvh", "chain")[int(rng.integers(0, 3))] if n_tri < 120 else ("sah", "lbvh")[int(rng.integers(0, 2))]
    bvh = os.environ.get("GLRT_FUZZ_BVH", bvh)  # (a soak under one builder, e.g. GLRT_FUZZ_BVH=reference: the seeds' own draws stay what they were)
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


def test_mutated_wire_scenes(gpu_device):
    """Wire-format scenes with a few floats mutated -- tree children and boxes, triangle indices and materials, lights, material parameters, vertices: NaN,
    +-inf, -1, 0.5, 2^24 + 1, 3e38, copies of other entries.  What glrtx_upload_scene accepts renders like the oracle (ray counts included, NaNs as NaNs); what
    it refuses is refused with GLRTX_ESCENE or GLRTX_EDEPTH.  (tools/gpu_wire_fuzz.py is the long form: 5600 mutations in round 4, 3892 accepted, 0 mismatches.)"""
    from oracle import pt_oracle
    from glrt_amd import device
    rng = np.random.default_rng(11)
    specials = np.array([np.nan, np.inf, -np.inf, -1.0, -0.0, 0.0, 0.5, 1.0, 2.0, 1e9, 1.7e7, 16777216.0, 16777217.0, 3e38, -3e38, 2147483648.0, 4294967296.0, 1e-40], np.float32)
    bases = []
    for kind in ("sah", "chain", "lbvh"):
        bases.append(scenes.config_c3(32, 24, n=37, bvh=kind, max_depth=3))
        bases.append(scenes.config_c1(32, 24, bvh=kind, subdiv=1, max_depth=3))
    d = gpu_device
    accepted = refused = 0
    for it in range(int(os.environ.get("GLRT_WIRE_FUZZ", "150"))):
        sc0, pr = bases[it % len(bases)]
        sc = dict(sc0)
        key = ("bvh", "bvh", "bvh", "tri", "light", "mat", "vert")[int(rng.integers(0, 7))]
        a = np.array(sc[key], np.float32).copy().reshape(-1)
        for _ in range(int(rng.integers(1, 6))):
            i, mode = int(rng.integers(0, a.size)), int(rng.integers(0, 4))
            if mode == 0:
                a[i] = specials[int(rng.integers(0, specials.size))]
            elif mode == 1:
                a[i] = a[int(rng.integers(0, a.size))]
            elif mode == 2:
                a[i] = float(rng.integers(-5, a.size))
            else:
                a[i] = a[i] + 1.0
        sc[key] = a.reshape(np.shape(sc[key]))
        try:
            d.upload_scene(sc)
        except device.GlrtxError as e:
            assert e.code in (device.GLRTX_ESCENE, device.GLRTX_EDEPTH), e
            refused += 1
            continue
        accepted += 1
        ref, ref_rays = pt_oracle.render(sc, pr)
        d.set_partition(0, 1, 16); d.resize(32, 24); d.clear(); d.count_rays(True); d.reset_stats(); d.render(pr); d.sync()
        acc = d.read_accum()
        same = (acc.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(acc) & np.isnan(ref))
        assert same.all(), f"mutation {it} of {key}: {int((~same).any(-1).sum())} pixels differ"
        assert d.stats().rays == ref_rays, f"mutation {it} of {key}"
    assert accepted > 50 and refused > 20, (accepted, refused)


def test_hostile_render_parameters(gpu_device):
    """Camera matrices with NaN / inf / zero / random entries, aperture and focal length NaN, inf, negative, zero, denormal, zero samples, zero depth, seeds at
    infinity: whatever the reference's arithmetic makes of them, bit for bit (NaNs as NaNs).  (tools/gpu_param_fuzz.py: 1500 sets in round 4, 0 mismatches.)"""
    from oracle import pt_oracle
    rng = np.random.default_rng(5)
    specials = [np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-30, 1e30, -1.0, 1e-45, 3e38]
    sc, pr0 = scenes.config_c1(40, 28, max_depth=3, n_samples=2, subdiv=1)
    d = gpu_device
    d.upload_scene(sc); d.set_partition(0, 1, 16); d.resize(40, 28)
    for it in range(100):
        p = dict(pr0)
        c2w, s2c = np.array(p["c2w"], np.float32).copy(), np.array(p["s2c"], np.float32).copy()
        for _ in range(int(rng.integers(0, 3))):
            m = c2w if rng.integers(0, 2) else s2c
            m[int(rng.integers(0, 16))] = specials[int(rng.integers(0, len(specials)))] if rng.integers(0, 2) else float(rng.normal()) * 10
        p["c2w"], p["s2c"] = c2w, s2c
        if rng.integers(0, 3) == 0:
            p["aperture"] = float(specials[int(rng.integers(0, len(specials)))])
        if rng.integers(0, 3) == 0:
            p["focal"] = float(specials[int(rng.integers(0, len(specials)))])
        if rng.integers(0, 5) == 0:
            p["n_samples"] = int(rng.integers(0, 4))
        if rng.integers(0, 5) == 0:
            p["max_depth"] = int(rng.integers(0, 4))
        if rng.integers(0, 4) == 0:
            p["seed"] = (float(specials[int(rng.integers(0, len(specials)))]), float(rng.uniform()))
        d.clear(); d.count_rays(True); d.reset_stats(); d.render(p); d.sync()
        ref, ref_rays = pt_oracle.render(sc, p)
        acc = d.read_accum()
        same = (acc.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(acc) & np.isnan(ref))
        assert same.all(), f"parameter set {it}: {int((~same).any(-1).sum())} pixels differ ({p})"
        assert d.stats().rays == ref_rays, f"parameter set {it}"


def test_resolve_on_hostile_accumulators(gpu_device):
    """The resolve kernel against the oracle's restatement of screen.frag on finite but hostile accumulators (denormals, negatives, 1e-30 .. 3e38, counts of 0, 3e38
    and fractions) with gammas from 1e-45 to 3e38 and inf: both flush denormals, as llvmpipe does (pinned by tests/golden/screen_extreme.npz).  NaN / infinite
    texels and negative zeros are outside the pass's domain (tools/gpu_resolve_fuzz.py: 600 buffers in round 4, 0 mismatches)."""
    import torch
    from oracle import pt_oracle
    rng = np.random.default_rng(21)
    specials = np.array([0.0, 1e-45, 1e-40, 1.1754944e-38, 1e-30, 1e-6, 0.5, 1.0, 1.0000001, 2.0, 100.0, 1e30, 3e38, -1.0, -1e-30], np.float32)
    d = gpu_device
    W = H = 64
    d.resize(W, H)
    try:
        for it in range(80):
            acc = rng.uniform(0, 4, (H, W, 4)).astype(np.float32)
            acc[..., 3] = rng.integers(0, 5, (H, W)).astype(np.float32)
            m = rng.uniform(0, 1, acc.shape) < 0.15
            acc[m] = specials[rng.integers(0, specials.size, int(m.sum()))]
            e = rng.uniform(0, 1, acc.shape) < 0.2
            with np.errstate(all="ignore"):
                acc[e] = (acc[e] * np.float32(2.0) ** rng.integers(-140, 120, int(e.sum())).astype(np.float32)).astype(np.float32)
            acc[~np.isfinite(acc)] = 1.0
            acc[(acc == 0) & np.signbit(acc)] = 0.0
            acc[(np.abs(acc) < 1.1754944e-38) & (acc < 0)] = 0.0
            gamma = float(rng.choice([2.2, 1.0, 0.45, 1e-20, 1e20, np.inf, 3.0, 1e-45, 3e38, float(rng.uniform(0.1, 5))]))
            t = torch.from_numpy(np.ascontiguousarray(acc)).cuda()
            d.bind_accum(t.data_ptr(), W * 16, H)
            got = d.resolve_rgba8(gamma=gamma, flip_y=False)
            with np.errstate(all="ignore"):
                want = pt_oracle.resolve(acc, gamma, flip_y=False)
            assert np.array_equal(got, want), f"buffer {it}, gamma {gamma}: {int((got != want).sum())} bytes differ"
    finally:
        d.bind_accum(0, 0, 0)


def test_odd_image_sizes_and_partitions(gpu_device):
    """Images from 1 x 1 to 4097 x 2 (and transposes): one launch per frame, frames in flight, and row-stripe partitions of 2 .. 5 ranks with stripes of 8 .. 40
    rows, stitched -- three accumulated frames against the oracle, ray counts included."""
    from glrt_amd import dist
    from oracle import pt_oracle
    d = gpu_device
    sc, _ = scenes.config_c1(8, 8, max_depth=3, n_samples=1, subdiv=1)
    d.upload_scene(sc)
    rng = np.random.default_rng(3)
    try:
        for (w, h) in [(1, 1), (1, 2), (2, 1), (3, 3), (7, 5), (9, 9), (63, 1), (65, 3), (127, 17), (1, 129), (1000, 3), (3, 1000), (4097, 2), (2, 4097), (257, 255)]:
            c2w, s2c = scenes.camera((0, 5, 16), (0, 2.0, 0), (0, 1, 0), 40.0, w, h)
            p = scenes.make_params(c2w, s2c, w, h, 3, 1, seed=(0.2, 0.7))
            seeds = [host.frame_seed(f) for f in range(3)]
            ref, rays = None, 0
            for sd in seeds:
                ref, n = pt_oracle.render(sc, dict(p, seed=sd), accum=ref)
                rays += n
            for mode in ("one launch per frame", "frames in flight"):
                d.set_partition(0, 1, 16); d.resize(w, h); d.clear(); d.count_rays(True); d.reset_stats()
                if mode == "frames in flight":
                    d.render_frames(p, seeds)
                else:
                    for sd in seeds:
                        d.render(dict(p, seed=sd))
                d.sync()
                assert_bit_equal(d.read_accum(), ref, f"{w}x{h}, {mode}")
                assert d.stats().rays == rays, f"{w}x{h}, {mode}"
            world, stripe = int(rng.integers(2, 6)), int(rng.choice([8, 16, 24, 40]))
            stitched, total = np.zeros_like(ref), 0
            for rank in range(world):
                d.set_partition(rank, world, stripe); d.resize(w, h); d.clear(); d.count_rays(True); d.reset_stats()
                d.render_frames(p, seeds); d.sync()
                ys = dist.owned_rows(rank, world, stripe, h)
                a = d.read_accum()
                assert a.shape[0] == len(ys)
                if len(ys):
                    stitched[ys] = a
                total += d.stats().rays
            assert_bit_equal(stitched, ref, f"{w}x{h}, {world} ranks, {stripe}-row stripes")
            assert total == rays
    finally:
        d.set_partition(0, 1, 16)


@pytest.mark.parametrize("seed", range(48))
def test_shadow_hostile_scenes_match_the_oracle(gpu_device, monkeypatch, seed):
    """tests/fuzz_scenes.py: shadow_hostile -- lights flush in their ancestors' boxes, slivers, grazing light samples, |coordinates| up to 1e7 -- on the device's
    DEFAULT search for shadow rays (the reference's own, round 5) against the oracle (which the live reference confirms on these scenes,
    tests/test_reference_live.py); every form of the node fetch, both compilations, and the megakernel."""
    from fuzz_scenes import shadow_hostile
    from oracle import pt_oracle
    tag, scene, params = shadow_hostile(seed)
    monkeypatch.setenv("GLRTX_PAIR_FETCH", "012"[seed % 3])
    ref, ref_rays = pt_oracle.render(scene, params)
    d = gpu_device
    d.upload_scene(scene); d.set_partition(0, 1, 16); d.resize(params["width"], params["height"])
    try:
        for variant, count in ((2, True), (2, False), (1, True)):
            d.set_variant(variant); d.clear(); d.reset_stats(); d.count_rays(count)
            d.render(params); d.sync()
            assert_bit_equal(d.read_accum(), ref, f"{tag}, variant {variant}, counting {count}")
            if count:
                assert d.stats().rays == ref_rays
    finally:
        d.set_variant(2); d.count_rays(True)
