"""Parity tests proper: the HIP path, called through the C ABI (libglrtx.so), against
 (1) the committed golden images from the reference's own shader (tests/golden),
 (2) the CPU oracle on seeded inputs at sizes it finishes in seconds,
 (3) size-independent properties at BASELINE.json's full sizes.
Bar: bit-exact float32 (stricter than north_star's 1e-4; the tolerance is asserted too)."""
import numpy as np
import pytest

from conftest import PKG, screen_cases, assert_bit_equal, golden_names, load_golden
from glrt_amd import device, dist, host, scenes

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star: "pixels within 1e-4 of the GL reference"

# The render kernels exist in two compilations: with ray counting (what the ray-count assertions need) and without (what bench.py
# TIMES: pt_render_wgwf<false, *>, its own register allocation around the hand-written inline asm).  The image tests run on both.
BOTH_INSTANTIATIONS = pytest.mark.parametrize("count_rays", [True, False], ids=["counting", "timed"])
# ... and, since round 4, with three forms of the traversal step's node fetch: one record per lane, pair-cooperative (the two lanes of a pair fetch one
# lane's record between them and exchange the pieces; the host picks it for large trees), and the two in alternate steps (small trees).  GLRTX_PAIR_FETCH
# forces one; the image tests run on all three.
BOTH_NODE_FETCHES = pytest.mark.parametrize("pair_fetch", ["0", "1", "2"], ids=["lane-fetch", "pair-fetch", "alternating-fetch"])


def gpu_render(d, scene, params, frames=None, count_rays=True):
    d.upload_scene(scene)
    d.set_partition(0, 1, 16)
    d.resize(params["width"], params["height"])
    d.reset_stats()
    d.count_rays(count_rays)
    for sd in (frames or [params["seed"]]):
        d.render(dict(params, seed=sd))
    d.sync()
    return d.read_accum(), d.stats()


@BOTH_NODE_FETCHES
@BOTH_INSTANTIATIONS
@pytest.mark.parametrize("name", golden_names())
def test_hip_matches_reference_golden(gpu_device, monkeypatch, name, count_rays, pair_fetch):
    monkeypatch.setenv("GLRTX_PAIR_FETCH", pair_fetch)
    scene, params, rows, frames, rgb, cnt = load_golden(name)
    acc, st = gpu_render(gpu_device, scene, params, frames, count_rays=count_rays)
    acc = acc[rows[0]:rows[1]]
    assert np.nanmax(np.abs(acc[..., :3] - rgb)) <= TOL
    assert_bit_equal(acc[..., :3], rgb, f"{name} rgb")
    assert_bit_equal(acc[..., 3], cnt, f"{name} count")


ORACLE_CASES = [
    ("c1", dict(width=256, height=256, max_depth=1, n_samples=1)),                  # BASELINE configs[0] at full size
    ("c1", dict(width=200, height=120, max_depth=16, n_samples=4)),
    ("c2", dict(width=480, height=270, max_depth=4, n_samples=1)),
    ("headline", dict(width=480, height=270)),
    ("c3", dict(width=240, height=135, max_depth=1, n=10_000)),                      # 10k triangles, chain BVH
    ("c3", dict(width=240, height=135, max_depth=1, n=10_000, bvh="sah")),
    ("c4", dict(width=384, height=216, max_depth=8, n_samples=16)),
    ("c5", dict(width=480, height=270, max_depth=4, n=100_000)),                     # 100k triangles
]


@pytest.mark.parametrize("cfg,kw", ORACLE_CASES)
def test_hip_bit_exact_vs_oracle(gpu_device, monkeypatch, cfg, kw):
    from oracle import pt_oracle
    scene, params = scenes.CONFIGS[cfg](**kw)
    ref, ref_rays = pt_oracle.render(scene, params)
    fetches = set()
    for pair_fetch in ("0", "1", "2", None):  # every form of the node fetch forced, then the host's own choice
        if pair_fetch is None:
            monkeypatch.delenv("GLRTX_PAIR_FETCH")
        else:
            monkeypatch.setenv("GLRTX_PAIR_FETCH", pair_fetch)
        acc, st = gpu_render(gpu_device, scene, params)
        fetches.add(st.node_fetch_last)
        assert st.rays == ref_rays
        assert_bit_equal(acc, ref, f"{cfg} {kw} pair_fetch={pair_fetch}")
        acc, st = gpu_render(gpu_device, scene, params, count_rays=False)  # the instantiation bench.py times
        assert st.rays == 0
        assert_bit_equal(acc, ref, f"{cfg} {kw}, kernel without ray counting, pair_fetch={pair_fetch}")
    if "chain" not in str(scene.get("bvh_kind", "")):
        assert fetches == {0, 1, 2}, "every form of the node fetch ran"
    if cfg == "c5":
        assert st.node_fetch_last == 1, "100 k triangles: the host picks the pair-cooperative fetch by itself"
    if cfg == "headline":
        assert st.node_fetch_last == 0, "10 k triangles: one record per lane (the alternating form until the path state moved to queue positions, round 5)"


def test_dof_and_seed_sweep_vs_oracle(gpu_device):
    from oracle import pt_oracle
    scene, params = scenes.config_c1(128, 96, max_depth=6, n_samples=2)
    rng = np.random.default_rng(11)
    for i in range(4):
        p = dict(params, seed=tuple(float(np.float32(v)) for v in rng.uniform(0, 1, 2)),
                 aperture=0.25 * (i % 2), focal=8.0)
        ref, _ = pt_oracle.render(scene, p)
        acc, _ = gpu_render(gpu_device, scene, p)
        assert_bit_equal(acc, ref, f"seed {p['seed']} aperture {p['aperture']}")


@pytest.mark.parametrize("kind", ["sah", "chain"])
def test_scaled_scenes_match_the_oracle(gpu_device, kind):
    """BASELINE config 1's scene scaled by powers of ten -- geometry, BVH boxes and camera alike.  Small scales drive products into the denormal range (flushed on
    both sides), large ones make an ulp of a distance exceed EPS: at 1000x the shadow rays' range limit, whose margin was the absolute 2 EPS until round 4, culled
    the boxes of lights lying flush in them (a box's computed entry distance a few ulps beyond the light's computed t) and one pixel of this image went dark."""
    from oracle import pt_oracle
    scene0, params0 = scenes.config_c1(48, 32, max_depth=4, n_samples=1, bvh=kind, subdiv=1)
    for k in (1e-12, 1e-6, 1e-3, 1e2, 1e3, 1e4, 1e6, 1e12):
        kf = np.float32(k)
        vert = scene0["vert"].reshape(-1, 5, 3).copy()
        vert[:, 0] *= kf
        nodes = scene0["bvh"].reshape(-1, 9).copy()
        nodes[:, 0:6] *= kf
        sc = dict(scene0, vert=vert.reshape(-1, 3), bvh=nodes.reshape(-1, 3))
        c2w = np.array(params0["c2w"], np.float32).reshape(4, 4).copy()
        c2w[3, :3] *= kf
        p = dict(params0, c2w=c2w.reshape(-1))
        ref, ref_rays = pt_oracle.render(sc, p)
        for count in (True, False):
            acc, st = gpu_render(gpu_device, sc, p, count_rays=count)
            if count:
                assert st.rays == ref_rays
            assert_bit_equal(acc, ref, f"{kind} tree, scale {k:g}, counting {count}")


@pytest.mark.parametrize("kind", ["sah", "chain"])
def test_scenes_far_from_the_origin_match_the_oracle(gpu_device, kind):
    """Config 2's scene (Cornell-style box, icospheres, ceiling emitter) moved 1e5 .. 3e6 units away from the origin, camera and all, the tree rebuilt around the
    rounded vertices: coordinates of which an ulp is 0.008 .. 0.25 with distances of ~10."""
    from oracle import pt_oracle
    scene0, params0 = scenes.config_c2(48, 32, 4, 1, kind, 1)
    for off in ((1e5, 1e5, -1e5), (3e6, 0.0, 1e6)):
        o = np.asarray(off, np.float32)
        vert = scene0["vert"].reshape(-1, 5, 3).copy()
        vert[:, 0] += o
        sc = scenes.rebuild_bvh(dict(scene0, vert=vert.reshape(-1, 3)), kind)
        c2w = np.array(params0["c2w"], np.float32).reshape(4, 4).copy()
        c2w[3, :3] += o
        p = dict(params0, c2w=c2w.reshape(-1))
        ref, ref_rays = pt_oracle.render(sc, p)
        acc, st = gpu_render(gpu_device, sc, p)
        assert st.rays == ref_rays
        assert_bit_equal(acc, ref, f"{kind} tree, offset {off}")


def test_extreme_material_parameters(gpu_device):
    """Conductors of roughness 1e-8 .. 1e4 and eta / kappa of 0 .. 1e19, albedos of 0, 10, -1, 1e30 / 1e-40, inf and NaN, emitters of 1e30, 1e-40, negative,
    inf and NaN: whatever the reference's arithmetic makes of them (the radiance clamp swallows the non-finite values), bit for bit."""
    from oracle import pt_oracle
    from glrt_amd.scenes import SceneBuilder, quad, conductor, diffuse, emitter, camera, make_params
    W, H = 64, 48
    inf, nan = float("inf"), float("nan")
    grey = diffuse((0.7, 0.7, 0.7))
    cases = [(f"conductor alpha {a:g}", [grey, grey, conductor((0.2, 0.9, 1.1), (3.9, 2.4, 2.2), a), conductor((1.5,) * 3, (0.0,) * 3, a)], (10.0,) * 3) for a in (1e-8, 1e-2, 1e4)]
    cases += [(f"conductor eta {e:g} kappa {k:g}", [grey, conductor((e,) * 3, (k,) * 3, 0.1), conductor((e,) * 3, (k,) * 3, 0.5), grey], (10.0,) * 3)
              for e, k in ((0.0, 0.0), (1e3, 1e-3), (1e19, 1e19), (-1.0, 2.0))]
    cases += [(f"albedo {alb}", [diffuse(alb), grey, diffuse(alb), grey], (10.0,) * 3) for alb in ((0.0,) * 3, (10.0, 5.0, 1.0), (-1.0, 0.5, 2.0), (1e30, 1e-30, 1e-40), (inf, 0.5, 0.5), (nan, 0.5, 0.5))]
    cases += [(f"emitter {e}", [grey] * 4, e) for e in ((1e30,) * 3, (1e-30, 1e-40, 0.0), (-5.0, 1.0, 1.0), (inf, 1.0, 1.0), (nan, 1.0, 1.0))]
    for tag, mats, lamp_e in cases:
        b = SceneBuilder()
        ids = [b.add_material(m) for m in mats]
        lamp = b.add_material(emitter(lamp_e))
        b.add_mesh(*quad((-4, 0, 4), (8, 0, 0), (0, 0, -8)), ids[0])
        b.add_mesh(*quad((-4, 0, -4), (8, 0, 0), (0, 6, 0)), ids[1])
        b.add_mesh(*quad((-2.5, 0.01, 1.0), (2, 0, 0), (0, 2, -1)), ids[2])
        b.add_mesh(*quad((0.5, 0.01, 1.0), (2, 0, 0), (0, 2, -1)), ids[3])
        b.add_mesh(*quad((-1.5, 5.5, -1.5), (3, 0, 0), (0, 0, 3)), lamp)
        sc = b.build("sah")
        c2w, s2c = camera((0, 2.5, 8), (0, 1.0, 0), (0, 1, 0), 45.0, W, H)
        p = make_params(c2w, s2c, W, H, 6, 2, seed=(0.31, 0.62))
        ref, ref_rays = pt_oracle.render(sc, p)
        assert np.isfinite(ref).all(), f"{tag}: the reference's clamp keeps the accumulator finite"
        acc, st = gpu_render(gpu_device, sc, p)
        assert st.rays == ref_rays, tag
        assert_bit_equal(acc, ref, tag)


def test_seeds_far_outside_the_unit_interval(gpu_device):
    """u_seed is a pair of rand() values in [0, 1) in the reference (window.cpp:226-229), but the uniform takes any float: the hash's sin() then sees arguments
    up to 1e22 -- beyond the float -> int conversion's range, where the reference's GL implementation returns INT_MIN --, infinities and NaNs."""
    from oracle import pt_oracle
    scene, params = scenes.config_c1(64, 48, max_depth=4, n_samples=2)
    inf, nan = float("inf"), float("nan")
    for seed in [(-0.5, 1.5), (1234.5, -77.25), (1.0e6, -3.0e5), (2.0e7, 0.5), (3.0e9, 0.5), (1.0e20, -1.0e20), (inf, 0.5), (nan, 0.5)]:
        p = dict(params, seed=seed)
        ref, ref_rays = pt_oracle.render(scene, p)
        for count in (True, False):
            acc, st = gpu_render(gpu_device, scene, p, count_rays=count)
            if count:
                assert st.rays == ref_rays
            assert_bit_equal(acc, ref, f"seed {seed}, counting {count}")


# ---------------------------------------------------------------- full-size properties (no oracle at 1080p/4K)
def test_full_size_determinism_and_partition_invariance(gpu_device):
    """1920x1080, 8 bounces (the headline config): two runs agree bitwise, and stitching the row-stripe
    partitions of world 2 and 8 reproduces the single-partition image bitwise (global pixel coordinates)."""
    d = gpu_device
    scene, params = scenes.config_headline()
    full, st = gpu_render(d, scene, params)
    again, _ = gpu_render(d, scene, params)
    assert_bit_equal(full, again, "run-to-run")
    assert st.rays > 2 * 1920 * 1080 and np.all(full[..., 3] == 1.0)
    assert np.isfinite(full).all() and full[..., :3].max() <= 100.0 and full[..., :3].min() >= 0.0
    h = params["height"]
    for world, stripe in ((2, 16), (8, 8), (8, 24)):  # 8 rows = what bench.py and glrtx_group use
        stitched = np.zeros_like(full)
        total_rays = 0
        for rank in range(world):
            d.set_partition(rank, world, stripe)
            d.resize(params["width"], h)
            d.reset_stats()
            d.count_rays(True)
            d.render(params)
            d.sync()
            ys = dist.owned_rows(rank, world, stripe, h)
            assert np.array_equal(d.local_rows_y(), ys)
            stitched[ys] = d.read_accum()
            total_rays += d.stats().rays
        assert_bit_equal(stitched, full, f"world {world}, {stripe}-row stripes")
        assert total_rays == st.rays
    d.set_partition(0, 1, 16)


def test_accumulation_is_additive_over_frames(gpu_device):
    """Frames accumulate by read-modify-write: count == number of frames, and the sum of two single-frame
    images equals the two-frame accumulator up to one rounding per add (exactly: a + b in float32)."""
    d = gpu_device
    scene, params = scenes.config_c2(640, 360, max_depth=4)
    s0, s1 = host.frame_seed(0), host.frame_seed(1)
    a, _ = gpu_render(d, scene, dict(params, seed=s0))
    b, _ = gpu_render(d, scene, dict(params, seed=s1))
    both, _ = gpu_render(d, scene, params, frames=[s0, s1])
    assert np.all(both[..., 3] == 2.0)
    assert_bit_equal(both[..., :3], a[..., :3] + b[..., :3], "two-frame accumulation")
    assert not np.array_equal(a, b)


def test_4k_16spp_one_pass_equals_16_counts(gpu_device):
    """Config 4 as BASELINE names it: 3840x2160, 8 bounces, 16 spp in ONE pass (132.7 M paths in one launch, chunked by the memory budget).  Properties only here;
    tests/test_gpu_fullsize.py compares row bands of the same render with the oracle."""
    d = gpu_device
    scene, params = scenes.config_c4(3840, 2160, max_depth=8, n_samples=16)
    acc, st = gpu_render(d, scene, params)
    assert acc.shape == (2160, 3840, 4) and np.all(acc[..., 3] == 16.0)
    assert st.paths == 3840 * 2160 * 16 and st.rays >= st.paths
    assert np.isfinite(acc).all() and acc[..., :3].max() <= 1600.0


def test_resolve_rgba8_byte_exact_on_a_render(gpu_device):
    from oracle import pt_oracle
    """glrtx_resolve_rgba8 on a rendered image == the oracle's restatement of screen.frag + the RGBA8 read-back, byte for byte."""
    d = gpu_device
    scene, params = scenes.config_c1(256, 128, max_depth=4, n_samples=4)
    acc, _ = gpu_render(d, scene, params)
    img = d.resolve_rgba8(gamma=2.2, flip_y=True)
    assert img.shape == (128, 256, 4) and np.all(img[..., 3] == 255)
    assert np.array_equal(img, pt_oracle.resolve(acc, 2.2, flip_y=True))
    assert np.array_equal(d.resolve_rgba8(gamma=2.2, flip_y=False), img[::-1])


@pytest.mark.parametrize("case", screen_cases(), ids=lambda c: c[0])
def test_resolve_rgba8_reproduces_reference_bytes(gpu_device, case):
    """The HIP resolve kernel against the bytes the reference's own screen.frag produced on llvmpipe (tests/golden/screen_*.npz):
    every byte boundary of pow(x, 1/2.2) (sweep), random radiances / counts / gammas with NaN, zero-count and denormal texels, and a
    rendered image.  The accumulator contents are fed through a bound torch tensor."""
    import torch
    name, acc, gamma, want = case
    d = gpu_device
    h, w = acc.shape[:2]
    d.resize(w, h)
    t = torch.from_numpy(np.ascontiguousarray(acc)).cuda()
    d.bind_accum(t.data_ptr(), w * 16, h)
    try:
        got = d.resolve_rgba8(gamma=gamma, flip_y=False)
        assert np.array_equal(got, want), f"{name}: {int((got != want).sum())} bytes differ"
        assert np.array_equal(d.resolve_rgba8(gamma=gamma, flip_y=True), want[::-1])
    finally:
        d.bind_accum(0, 0, 0)


def test_bound_torch_accumulator_and_stream(gpu_device):
    """glrtx_bind_accum / glrtx_set_stream: render straight into a torch CUDA tensor on torch's stream
    (how bench.py hands rows to RCCL)."""
    import torch
    d = gpu_device
    scene, params = scenes.config_c1(160, 96, max_depth=3, n_samples=1)
    ref, _ = gpu_render(d, scene, params)
    t = torch.zeros((96, 160, 4), dtype=torch.float32, device="cuda")
    d.bind_accum(t.data_ptr(), 160 * 16, 96)
    d.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        d.render(params)
        d.sync()
        torch.cuda.synchronize()
        assert_bit_equal(t.cpu().numpy(), ref, "bound accumulator")
        # a shape that does not fit the caller's buffer is refused while it is bound (the ctx cannot grow it)
        for call in (lambda: d.resize(160, 97), lambda: d.resize(161, 96)):
            with pytest.raises(device.GlrtxError) as e:
                call()
            assert e.value.code == device.GLRTX_EINVAL and "bound accumulator" in str(e.value)
        with pytest.raises(device.GlrtxError):
            d.bind_accum(t.data_ptr(), 160 * 16, 95)  # fewer rows than the partition owns
        d.bind_accum(t.data_ptr(), 160 * 16, 96)
        d.set_partition(0, 2, 16)  # half the rows: fits
        d.set_partition(0, 1, 16)
    finally:
        d.set_stream(0)
        d.bind_accum(0, 0, 0)


def test_render_launches_are_ordered_with_torch_work_on_the_same_stream(gpu_device):
    """What bench.py relies on for the RCCL gather: with a torch stream of its own set on the context, a torch kernel enqueued right after
    glrtx_render_frames -- no sync in between -- sees the finished accumulator, and a zero_() enqueued before a launch is seen by it.
    (torch's default stream has handle 0, which glrtx_set_stream takes as "the context's own stream": no ordering with torch there.)"""
    import torch
    d = gpu_device
    scene, params = scenes.config_headline()
    W, H = params["width"], params["height"]
    seeds = [host.frame_seed(f) for f in range(4)]
    ref, _ = gpu_render(d, scene, params, frames=seeds, count_rays=False)
    stream = torch.cuda.Stream()
    assert stream.cuda_stream != 0
    t = torch.full((H, W, 4), 7.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    d.bind_accum(t.data_ptr(), W * 16, H)
    d.set_stream(stream.cuda_stream)
    try:
        with torch.cuda.stream(stream):
            t.zero_()                                   # enqueued, not waited for
            d.render_frames(params, seeds)              # 4 x 1080p: a few milliseconds of device time
            snap = t.clone()                            # enqueued right behind the launch
            total = t.sum(dtype=torch.float64)
        stream.synchronize()
        assert_bit_equal(snap.cpu().numpy(), ref, "clone enqueued behind the launch")
        assert abs(float(total) - float(ref.astype(np.float64).sum())) <= 1e-6 * abs(float(ref.astype(np.float64).sum()))
    finally:
        d.sync()
        d.set_stream(0)
        d.bind_accum(0, 0, 0)


def test_error_behaviour(gpu_device):
    d2 = device.Device()
    try:
        with pytest.raises(device.GlrtxError) as e:
            d2.render(scenes.config_c1(32, 32, subdiv=1)[1])
        assert e.value.code == device.GLRTX_EINVAL and "no scene" in str(e.value)
        scene, params = scenes.config_c1(32, 32, subdiv=1)
        bad = dict(scene, tri=scene["tri"].copy())
        bad["tri"][0, 0] = -5
        with pytest.raises(device.GlrtxError) as e:
            d2.upload_scene(bad)
        assert e.value.code == device.GLRTX_ESCENE
        d2.upload_scene(scene)
        with pytest.raises(device.GlrtxError):
            d2.render(params)  # no resize yet
        with pytest.raises(device.GlrtxError):
            d2.set_partition(2, 2, 16)
        with pytest.raises(device.GlrtxError):
            d2.set_partition(0, 2, 10)  # not a multiple of the tile height
    finally:
        d2.close()


def test_empty_and_edge_scenes(gpu_device):
    d = gpu_device
    empty = dict(vert=np.zeros((0, 3), np.float32), tri=np.zeros((0, 4), np.float32), mat=np.zeros((6, 3), np.float32),
                 light=np.zeros((0, 4), np.float32), bvh=np.zeros((0, 3), np.float32))
    _, params = scenes.config_c1(33, 17, subdiv=1)
    acc, st = gpu_render(d, empty, params)
    assert acc.shape == (17, 33, 4) and np.all(acc[..., :3] == 0) and np.all(acc[..., 3] == 1)
    assert st.rays == 33 * 17
    scene, params = scenes.config_c1(17, 9, max_depth=0, subdiv=1)  # max_depth 0: no rays, count still advances
    acc, st = gpu_render(d, scene, params)
    assert st.rays == 0 and np.all(acc[..., 3] == 1) and np.all(acc[..., :3] == 0)


def test_forks_with_an_absent_child_match_the_oracle(gpu_device):
    """The wire format lets a fork leave out children.x or children.y (raytrace.frag:299-307 pushes only indices >= 0); no builder here
    produces that.  The kernel refers such a child to a never-hit record instead of testing for its absence: same image, same ray count,
    in every kernel variant."""
    from oracle import pt_oracle
    scene, params = scenes.config_c1(96, 64, max_depth=5, n_samples=2, subdiv=1)
    nodes = scene["bvh"].reshape(-1, 9).copy()
    rng = np.random.default_rng(5)
    forks = np.flatnonzero(nodes[:, 8] < 0)
    extra = []
    for k, f in enumerate(rng.choice(forks, 40, replace=False)):
        side = 6 + (k & 1)                       # the child that gets a one-child fork put in front of it
        child = int(nodes[f, side])
        unary = nodes[child].copy()              # same box as the child it leads to
        unary[6:9] = (-1.0, -1.0, -1.0)
        unary[6 + ((k >> 1) & 1)] = float(child)  # ... hanging off children.x for some, children.y for others
        nodes[f, side] = float(nodes.shape[0] + len(extra))
        extra.append(unary)
    sc = dict(scene, bvh=np.concatenate([nodes, np.array(extra, np.float32)], 0).reshape(-1, 3))
    ref, ref_rays = pt_oracle.render(sc, params)
    d = gpu_device
    try:
        for v in (2, 1, 0):
            d.set_variant(v)
            acc, st = gpu_render(d, sc, params)
            assert st.rays == ref_rays
            assert_bit_equal(acc, ref, f"absent children, variant {v}")
    finally:
        d.set_variant(2)


@pytest.mark.parametrize("suspend_max", [0, 8, 24, 64])
def test_parked_rays_do_not_change_the_image(gpu_device, monkeypatch, suspend_max):
    """At the end of a trip a wave parks its last few path rays instead of running them alone, and the shade phase defers their paths
    (pt_kernel.hip.h: kSuspendMax).  Only the schedule changes: never (0), rarely, by default, and whenever the queue is empty (64) give
    the oracle's image and ray count, one launch per frame and frames in flight, in both compilations."""
    from oracle import pt_oracle
    monkeypatch.setenv("GLRTX_SUSPEND_MAX", str(suspend_max))
    d = gpu_device
    scene, params = scenes.CONFIGS["headline"](width=320, height=180)
    seeds = _seeds(3)
    ref, ref_rays = None, 0
    for sd in seeds:
        ref, n = pt_oracle.render(scene, dict(params, seed=sd), accum=ref)
        ref_rays += n
    for count in (True, False):
        acc, st = gpu_render(d, scene, params, frames=seeds, count_rays=count)
        assert_bit_equal(acc, ref, f"suspend_max {suspend_max}, consecutive launches, counting {count}")
        assert not count or st.rays == ref_rays
        d.clear(); d.reset_stats(); d.count_rays(count)
        d.render_frames(params, seeds); d.sync()
        assert_bit_equal(d.read_accum(), ref, f"suspend_max {suspend_max}, frames in flight, counting {count}")
        assert not count or d.stats().rays == ref_rays


# ---------------------------------------------------------------- the list scan (BASELINE config 3's brute-force "chain" tree; csrc/scan_asm.hip.h)
def _scan_case(d, sc, params, what, ref=None, ref_rays=None):
    from oracle import pt_oracle
    if ref is None:
        ref, ref_rays = pt_oracle.render(sc, params)
    for count in (True, False):
        acc, st = gpu_render(d, sc, params, count_rays=count)
        if count:
            assert st.rays == ref_rays
        assert_bit_equal(acc, ref, f"{what}, counting {count}")


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 14, 19, 64, 65, 130])
def test_list_scan_of_any_length(gpu_device, n):
    """The hand-written scan works through the list in groups of four records with the last leaf's record behind them (never-hit records fill the last
    group): every remainder of (n - 1) mod 4, the shortest list there is, and lists of about one and two wave's worth of records."""
    scene, params = scenes.config_c3(64, 48, max_depth=3, n=n, bvh="chain")
    _scan_case(gpu_device, scene, params, f"chain of {n}")


@pytest.mark.parametrize("n", [3, 4, 5, 6, 7, 9, 200])
def test_list_scan_with_a_box_per_fork(gpu_device, n):
    """A vine whose forks carry boxes of their own (each the bounds of the triangles still to come) takes the C++ statement of the scan, every
    fork's box tested against the running tHit; the chain builder's vine -- one box for all -- takes the hand-written one.  Same image: the boxes only cull."""
    from oracle import pt_oracle
    w, h = (96, 64) if n > 50 else (48, 32)
    scene, params = scenes.config_c3(w, h, max_depth=3, n=n, bvh="chain")
    ref, ref_rays = pt_oracle.render(scene, params)
    nodes = scene["bvh"].reshape(-1, 9).copy()
    leaf_of = lambda i: 2 * i + 1 if i < n - 1 else 2 * n - 2
    lo, hi = nodes[leaf_of(n - 1), 0:3].copy(), nodes[leaf_of(n - 1), 3:6].copy()
    for i in range(n - 2, -1, -1):  # fork i = node 2 i: bounds of leaves i .. n - 1
        lo = np.minimum(lo, nodes[leaf_of(i), 0:3]); hi = np.maximum(hi, nodes[leaf_of(i), 3:6])
        nodes[2 * i, 0:3], nodes[2 * i, 3:6] = lo, hi
    assert not np.array_equal(nodes[0, 0:6], nodes[2 * (n - 2), 0:6]), "the forks' boxes differ: not the one-box list"
    sc = dict(scene, bvh=nodes.reshape(-1, 3))
    ref2, rays2 = pt_oracle.render(sc, params)
    assert_bit_equal(ref2, ref, "oracle: tighter fork boxes do not change the image")
    _scan_case(gpu_device, sc, params, f"vine of {n} with a box per fork", ref2, rays2)
    _scan_case(gpu_device, scene, params, f"vine of {n} with one box", ref, ref_rays)


def test_huge_determinants_flush_like_the_reference(gpu_device):
    """Beyond |det| = 2^126 the reciprocal is a denormal: the reference's GL implementation flushes it to zero (no hit), and so must the device -- the
    scene that showed, in round 4, that the kernels had kept denormals since round 1 (684 of 1536 pixels differed).  1 / det is v_rcp_f32 + one Newton
    step, which in that float mode is the IEEE quotient of every normal det (test_short_quotients_equal_the_ieee_quotient_on_every_float).  Triangles
    with edges of ~1e19 and one corner in front of the camera have determinants of 4e37 .. 1.7e38 -- on both sides of 2^126 = 8.5e37, below FLT_MAX --
    and are hit next to that corner at t ~ 2 (u, v ~ 1e-19): the image is finite and lit, and it is the oracle's."""
    from oracle import pt_oracle
    b = scenes.SceneBuilder()
    grey = b.add_material(scenes.diffuse((0.7, 0.6, 0.5)))
    lamp = b.add_material(scenes.emitter((8.0, 8.0, 8.0)))
    E, pos, rng = 1.3e19, [], np.random.default_rng(5)
    for i in range(9):
        v0 = np.array([-0.6 + 0.1 * i, -0.5 + 0.07 * i, -2.0 - 0.2 * i])
        k = rng.uniform(0.5, 1.0, 2)
        pos.append([v0, v0 + [E * k[0], 0.0, -0.1 * E * (i % 3)], v0 + [0.0, E * k[1], 0.05 * E * (i % 2)]])
    b.add_mesh(np.array(pos), np.array([[[0, 0, 1]] * 3] * 9), grey)
    b.add_mesh(np.array([[[-1.5, 1.0, -1.0], [-1.0, 1.0, -1.0], [-1.5, 1.0, -1.6]], [[-1.0, 1.0, -1.0], [-1.0, 1.0, -1.6], [-1.5, 1.0, -1.6]]]),
               np.array([[[0, -1, 0]] * 3] * 2), lamp)
    c2w, s2c = scenes.camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 60.0, 48, 32, 0.1, 100.0)
    params = scenes.make_params(c2w, s2c, 48, 32, 3, 2)
    for kind in ("chain", "sah"):  # the list scan and the tree traversal
        sc = b.build(kind)
        ref, ref_rays = pt_oracle.render(sc, params)
        assert np.isfinite(ref).all() and (ref[..., :3].sum(-1) > 0).mean() > 0.1, "the giant triangles are seen and lit"
        _scan_case(gpu_device, sc, params, f"huge determinants, {kind} tree", ref, ref_rays)


@pytest.mark.parametrize("n", [12, 40, 63])
def test_deep_traversal_stacks_match_the_oracle(gpu_device, n):
    """A comb that stacks one entry per level (every fork = a leaf as children.x, the rest of the tree as children.y): the per-lane LDS
    stack of the hand-written step at depths up to the reference's own limit (int stack[64], raytrace.frag:284), in all three kernel forms."""
    from oracle import pt_oracle
    scene, params = scenes.config_c3(64, 48, max_depth=3, n=n, bvh="chain")
    nodes = scene["bvh"].reshape(-1, 9).copy()
    fk = nodes[:, 8] < 0
    nodes[fk, 6], nodes[fk, 7] = nodes[fk, 7].copy(), nodes[fk, 6].copy()
    sc = dict(scene, bvh=nodes.reshape(-1, 3))
    ref, ref_rays = pt_oracle.render(sc, params)
    d = gpu_device
    try:
        for v in (2, 1, 0):
            d.set_variant(v)
            for count in (True, False):
                acc, st = gpu_render(d, sc, params, count_rays=count)
                assert st.stack_entries == n - 2  # n - 1 forks, the last one a leaf pair
                if count:
                    assert st.rays == ref_rays
                assert_bit_equal(acc, ref, f"comb of {n}, variant {v}, counting {count}")
    finally:
        d.set_variant(2)


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_all_kernel_variants_bit_identical(gpu_device, variant):
    """The tile megakernel (0), the persistent megakernel with path regeneration (1) and the
    workgroup-local wavefront (2, default) share the shading/traversal code and must agree bitwise,
    including at sizes smaller than one workgroup block and with several samples per pass."""
    from oracle import pt_oracle
    d = gpu_device
    try:
        d.set_variant(variant)
        for cfg, kw in (("c1", dict(width=40, height=24, max_depth=5, n_samples=3, subdiv=1)),
                        ("c2", dict(width=320, height=180, max_depth=8, n_samples=2, subdiv=2)),
                        ("c3", dict(width=160, height=90, max_depth=2, n=3000))):
            scene, params = scenes.CONFIGS[cfg](**kw)
            ref, ref_rays = pt_oracle.render(scene, params)
            acc, st = gpu_render(d, scene, params)
            assert st.rays == ref_rays
            assert_bit_equal(acc, ref, f"variant {variant} {cfg}")
    finally:
        d.set_variant(2)


def _seeds(n, start=0):
    return [host.frame_seed(start + f) for f in range(n)]


@pytest.mark.parametrize("cfg,kw,n_frames", [
    ("c1", dict(width=50, height=38, max_depth=5, n_samples=1, subdiv=1), 3),    # NPOT, smaller than one workgroup block
    ("c1", dict(width=96, height=64, max_depth=4, n_samples=3, subdiv=1), 4),    # several samples per frame: planes = frames x samples
    ("c2", dict(width=320, height=180, max_depth=8, n_samples=1, subdiv=2), 5),
    ("c2", dict(width=64, height=64, max_depth=0, n_samples=2, subdiv=1), 3),    # u_maxDepth 0: samples finish as they start
])
def test_frames_in_flight_equal_consecutive_frames_and_oracle(gpu_device, cfg, kw, n_frames):
    """glrtx_render_frames(n) == n consecutive glrtx_render calls == the oracle run frame after frame, bitwise,
    including the ray count."""
    from oracle import pt_oracle
    d = gpu_device
    scene, params = scenes.CONFIGS[cfg](**kw)
    seeds = _seeds(n_frames)
    seq, st_seq = gpu_render(d, scene, params, frames=seeds)
    d.clear(); d.reset_stats(); d.count_rays(True)
    d.render_frames(params, seeds); d.sync()
    bat, st = d.read_accum(), d.stats()
    assert_bit_equal(bat, seq, f"{cfg} frames in flight vs consecutive launches")
    assert st.rays == st_seq.rays and st.launches == n_frames
    ref, ref_rays = None, 0
    for sd in seeds:
        ref, n = pt_oracle.render(scene, dict(params, seed=sd), accum=ref)
        ref_rays += n
    assert st.rays == ref_rays
    assert_bit_equal(bat, ref, f"{cfg} frames in flight vs oracle")
    # the instantiation bench.py times (no ray counting), in flight and frame by frame
    d.clear(); d.reset_stats(); d.count_rays(False)
    d.render_frames(params, seeds); d.sync()
    assert_bit_equal(d.read_accum(), ref, f"{cfg} frames in flight vs oracle, kernel without ray counting")
    assert d.stats().rays == 0
    seq_clean, _ = gpu_render(d, scene, params, frames=seeds, count_rays=False)
    assert_bit_equal(seq_clean, ref, f"{cfg} consecutive launches vs oracle, kernel without ray counting")


def test_frames_in_flight_full_size_partitions_and_accumulate_on_top(gpu_device):
    """Headline size: 8 frames in one launch on top of an already accumulated frame, whole image and as a 3-rank
    partition; both must equal 9 consecutive launches."""
    d = gpu_device
    scene, params = scenes.config_headline()
    seeds = _seeds(9)
    seq, _ = gpu_render(d, scene, params, frames=seeds, count_rays=False)
    d.clear(); d.render(dict(params, seed=seeds[0])); d.render_frames(params, seeds[1:]); d.sync()
    assert_bit_equal(d.read_accum(), seq, "1 + 8 frames in flight")
    assert np.all(seq[..., 3] == 9.0)
    try:
        stitched = np.zeros_like(seq)
        for rank in range(3):
            d.set_partition(rank, 3, 8); d.resize(params["width"], params["height"])
            d.render_frames(params, seeds); d.sync()
            stitched[d.local_rows_y()] = d.read_accum()
        assert_bit_equal(stitched, seq, "frames in flight, 3-rank partition")
    finally:
        d.set_partition(0, 1, 16)


def test_frames_in_flight_other_variants_fall_back_to_consecutive_launches(gpu_device):
    d = gpu_device
    scene, params = scenes.config_c1(width=64, height=48, max_depth=3, subdiv=1)
    seeds = _seeds(3)
    want, _ = gpu_render(d, scene, params, frames=seeds)
    try:
        for v in (0, 1):
            d.set_variant(v); d.clear(); d.render_frames(params, seeds); d.sync()
            assert_bit_equal(d.read_accum(), want, f"variant {v} render_frames")
    finally:
        d.set_variant(2)


@pytest.mark.parametrize("cfg,kw", [("c3", dict(n=1)), ("c3", dict(n=2)), ("c3", dict(n=777)), ("c3", dict(n=10_000)),
                                    ("c2", dict(subdiv=2)), ("c5", dict(n=100_000))])
def test_gpu_lbvh_equals_cpu_lbvh_bitwise(gpu_device, cfg, kw):
    """glrtx_build_lbvh (Morton sort + Karras hierarchy + bottom-up fit on the device) returns exactly the nodes of the
    CPU statement glrt_bvh_build_lbvh, and the same depth."""
    scene, _ = scenes.CONFIGS[cfg](width=16, height=16, bvh="lbvh", **kw)
    nodes, depth, ms = gpu_device.build_lbvh(scene["vert"], scene["tri"])
    want = np.asarray(scene["bvh_builder"], np.float32).reshape(-1, 3)  # (the builder's own output: scene["bvh"] has the light side first)
    assert nodes.shape == want.shape and depth == scene["bvh_depth"]
    assert_bit_equal(nodes, want, f"{cfg} LBVH nodes")
    assert ms > 0.0


@pytest.mark.parametrize("cfg,kw", [("c3", dict(n=1)), ("c3", dict(n=2)), ("c3", dict(n=3)), ("c3", dict(n=64)), ("c3", dict(n=65)), ("c3", dict(n=66)), ("c3", dict(n=129)),
                                    ("c3", dict(n=777)), ("c3", dict(n=10_000)), ("c2", dict(subdiv=2)), ("headline", dict()), ("c5", dict(n=100_000))])
def test_gpu_sah_by_levels_equals_the_cpu_statement_bitwise(gpu_device, cfg, kw):
    """glrtx_build_bvh_sah (round 5, csrc/sahl.hip.h: binned SAH level by level on the device -- segment bounds and bins by atomics on ordered keys, a thread per
    segment choosing its split, a scan numbering the children -- then the exact sweep SAH for the subtrees of <= 64 triangles) returns exactly the nodes of its CPU
    statement glrt_bvh_build_sah_levels, and the same depth; sizes around the 64-leaf boundary included."""
    scene, _ = scenes.CONFIGS[cfg](width=16, height=16, bvh="sahl", **kw)
    nodes, depth, ms = gpu_device.build_bvh_sah(scene["vert"], scene["tri"])
    want = np.asarray(scene["bvh_builder"], np.float32).reshape(-1, 3)  # (the builder's own output: scene["bvh"] has the light side first)
    assert nodes.shape == want.shape and depth == scene["bvh_depth"]
    assert_bit_equal(nodes, want, f"{cfg} SAH-by-levels nodes")
    assert ms > 0.0


def test_gpu_sah_by_levels_on_equal_centres_non_finite_vertices_and_a_bad_index(gpu_device):
    """Every open segment split by position (500 identical triangles); vertices at infinity / NaN / 3e38; a vertex index out of range."""
    from oracle import pt_oracle
    sc, _ = scenes.config_c3(8, 8, n=3, bvh="sah")
    v = np.tile(sc["vert"].reshape(-1, 5, 3)[:3], (500, 1, 1)).reshape(-1, 3)
    t = np.array([[3 * i, 3 * i + 1, 3 * i + 2, 0] for i in range(500)], np.float32)
    nodes, depth, _ = gpu_device.build_bvh_sah(v, t)
    want, want_depth = host.build_bvh(v, t, "sahl")
    assert_bit_equal(nodes, want, "equal centres"); assert depth == want_depth <= 14
    scene0, params = scenes.config_c1(64, 48, max_depth=4, n_samples=1, bvh="sah", subdiv=1)
    for tag, sc0 in _non_finite_variants(scene0):
        nodes, depth, _ = gpu_device.build_bvh_sah(sc0["vert"], sc0["tri"])
        want, want_depth = host.build_bvh(sc0["vert"], sc0["tri"], "sahl")
        assert depth == want_depth
        assert_bit_equal(nodes.view(np.uint32), np.asarray(want, np.float32).reshape(-1, 3).view(np.uint32), f"{tag}: SAH-by-levels nodes, device against CPU statement")
        s2 = dict(sc0, bvh=nodes, bvh_depth=depth, bvh_kind="sahl")
        ref, ref_rays = pt_oracle.render(s2, params)
        acc, st = gpu_render(gpu_device, s2, params)
        assert st.rays == ref_rays, tag
        assert_bit_equal(acc, ref, f"{tag} vertices, device-built SAH-by-levels tree")
    t[7, 1] = 1e6
    with pytest.raises(device.GlrtxError) as e:
        gpu_device.build_bvh_sah(v, t)
    assert e.value.code == device.GLRTX_ESCENE


def test_render_with_the_device_built_sah_tree_matches_the_oracle(gpu_device):
    """Config 5 at reduced resolution under the tree glrtx_build_bvh_sah builds: image and ray count equal the oracle's on that tree, and the image equals the
    CPU SAH tree's (random triangles: no exact ties)."""
    from oracle import pt_oracle
    d = gpu_device
    scene, params = scenes.config_c5(width=160, height=90, max_depth=4, bvh="sah")
    nodes, depth, _ = d.build_bvh_sah(scene["vert"], scene["tri"])
    sl = dict(scene, bvh=nodes, bvh_depth=depth, bvh_kind="sahl")
    acc, st = gpu_render(d, sl, params)
    ref, ref_rays = pt_oracle.render(sl, params)
    assert st.rays == ref_rays and st.node_fetch_last == 1
    assert_bit_equal(acc, ref, "c5 with the device-built SAH-by-levels tree")
    ref_sah, _ = pt_oracle.render(scene, params)
    assert_bit_equal(acc, ref_sah, "SAH-by-levels image vs SAH image")


@pytest.mark.parametrize("cfg,kw", [("c5", dict(width=160, height=90, max_depth=4)), ("c2", dict(width=128, height=72))])
def test_render_with_the_reinserted_tree_matches_the_oracle(gpu_device, cfg, kw):
    """The "sah-reinsert" builder (CPU SAH + glrt_bvh_reinsert + lights first): image and ray count equal the oracle's on that tree, and the image equals the builder
    tree's (these scenes have no exact ties)."""
    from oracle import pt_oracle
    scene, params = scenes.CONFIGS[cfg](**kw)
    sr = scenes.rebuild_bvh(scene, "sah-reinsert")
    assert not np.array_equal(np.asarray(sr["bvh"]), np.asarray(scene["bvh"]))
    acc, st = gpu_render(gpu_device, sr, params)
    ref, ref_rays = pt_oracle.render(sr, params)
    assert st.rays == ref_rays
    assert_bit_equal(acc, ref, f"{cfg} with the reinserted tree")
    assert_bit_equal(acc, pt_oracle.render(scene, params)[0], "reinserted-tree image vs builder-tree image")


def test_gpu_lbvh_many_equal_centres_and_bad_index(gpu_device):
    sc, _ = scenes.config_c3(8, 8, n=3, bvh="sah")
    v = np.tile(sc["vert"].reshape(-1, 5, 3)[:3], (500, 1, 1)).reshape(-1, 3)          # 500 identical triangles
    t = np.array([[3 * i, 3 * i + 1, 3 * i + 2, 0] for i in range(500)], np.float32)
    nodes, depth, _ = gpu_device.build_lbvh(v, t)
    want, want_depth = host.build_bvh(v, t, "lbvh")
    assert_bit_equal(nodes, want, "equal Morton codes"); assert depth == want_depth <= 10
    t[7, 1] = 1e6
    with pytest.raises(device.GlrtxError) as e:
        gpu_device.build_lbvh(v, t)
    assert e.value.code == device.GLRTX_ESCENE


def _non_finite_variants(scene):
    """The scene with a few vertex coordinates replaced: NaN, +inf, -inf, 3e38, and a mix (a triangle reaching from -inf to +inf has a NaN centre)."""
    for tag, val in (("nan", np.nan), ("+inf", np.inf), ("-inf", -np.inf), ("3e38", 3e38), ("mixed", None)):
        vert = scene["vert"].reshape(-1, 5, 3).copy()
        if val is None:
            vert[7, 0, 1], vert[8, 0, 1], vert[100, 0, 0], vert[101, 0, 0], vert[333, 0, 2] = np.inf, -np.inf, 3e38, -3e38, np.nan
        else:
            vert[7, 0, 1] = vert[100, 0, 0] = vert[333, 0, 2] = val
        yield tag, dict(scene, vert=vert.reshape(-1, 3))


def test_non_finite_vertices(gpu_device):
    """Vertices at infinity, at 3e38, NaN: the host builders take them (a box's centre that is not finite is ordered and binned as 0; until round 4 the SAH
    builder indexed its bins with (int)NaN and crashed), the device builds the same linear BVH as the CPU statement bit for bit, and the image is the oracle's
    under every tree."""
    from oracle import pt_oracle
    scene0, params = scenes.config_c1(64, 48, max_depth=4, n_samples=1, bvh="sah", subdiv=1)
    for tag, sc0 in _non_finite_variants(scene0):
        nodes, depth, _ = gpu_device.build_lbvh(sc0["vert"], sc0["tri"])
        want, want_depth = host.build_bvh(sc0["vert"], sc0["tri"], "lbvh")
        assert depth == want_depth
        assert_bit_equal(nodes.view(np.uint32), np.asarray(want, np.float32).reshape(-1, 3).view(np.uint32), f"{tag}: LBVH nodes, device against CPU statement")
        for kind in ("sah", "chain", "lbvh"):
            sc = scenes.rebuild_bvh(sc0, kind)
            ref, ref_rays = pt_oracle.render(sc, params)
            acc, st = gpu_render(gpu_device, sc, params)
            assert st.rays == ref_rays, (tag, kind)
            assert_bit_equal(acc, ref, f"{tag} vertices, {kind} tree")


def test_render_with_gpu_built_lbvh_matches_oracle(gpu_device):
    """BASELINE config 5 (100k triangles, linear BVH) at reduced resolution: tree from the GPU builder, image vs the oracle
    with the same tree -- and vs the oracle with the SAH tree (random triangles: no exact ties)."""
    from oracle import pt_oracle
    d = gpu_device
    scene, params = scenes.config_c5(width=160, height=90, max_depth=4, bvh="sah")
    nodes, depth, _ = d.build_lbvh(scene["vert"], scene["tri"])
    lb = dict(scene, bvh=nodes, bvh_depth=depth, bvh_kind="lbvh")
    acc, st = gpu_render(d, lb, params)
    ref, ref_rays = pt_oracle.render(lb, params)
    assert st.rays == ref_rays
    assert_bit_equal(acc, ref, "c5 with GPU-built LBVH")
    ref_sah, _ = pt_oracle.render(scene, params)
    assert_bit_equal(acc, ref_sah, "LBVH image vs SAH image")


def test_frames_in_flight_are_chunked_by_the_memory_budget(gpu_device, monkeypatch):
    """A request that does not fit the device-memory budget is issued as several launches, with the same result."""
    d = gpu_device
    scene, params = scenes.config_c2(width=256, height=144, max_depth=4, subdiv=1)
    seeds = _seeds(7)
    want, _ = gpu_render(d, scene, params, frames=seeds)
    per_frame_mb = 256 * 144 * 16 / 2**20   # one float4 sample plane per frame (path state is addressed by workgroup and queue position: constant, not charged)
    assert per_frame_mb == 0.5625
    # (round 6: launches on the context's own stream are fed launches, and up to three of them share the budget -- one renders, one drains, one is being fed: a third each)
    # 3 MB: one frame per launch; 6 MB: three fit, so 7 frames go as 3 + 3 + 1; 9 MB: five fit -> equal helpings of 4 + 3; 16 GB: one launch
    for budget_mb, launches in ((3, 7), (6, 3), (9, 2), (1 << 14, 1)):
        monkeypatch.setenv("GLRTX_FRAMES_BUDGET_MB", str(budget_mb))
        d.clear(); d.reset_stats()
        d.render_frames(params, seeds); d.sync()
        st = d.stats()
        assert st.launches == 7 and st.kernel_launches == launches, (budget_mb, st.kernel_launches)
        assert_bit_equal(d.read_accum(), want, f"budget {budget_mb} MB")


def test_path_state_does_not_grow_with_the_frames_in_flight(gpu_device):
    """Round 5: the wavefront kernel's path state is addressed by workgroup and path-queue position -- glrtx_stats.wf_state_mib is what the last launch ran on: the same for
    2 and for 24 frames in flight (two sets of six float4 planes for every workgroup slot of the device x 4096 paths: 3 MiB per CU), and both launches render what
    consecutive single-frame launches render."""
    d = gpu_device
    scene, params = scenes.config_c2(width=1280, height=720, max_depth=5, subdiv=1)  # (large enough for a full grid: the state is sized by the launch's workgroups, ABI 10)
    sizes = []
    for n in (2, 24):
        seeds = _seeds(n)
        want, _ = gpu_render(d, scene, params, frames=seeds)  # (uploads the scene; one launch per frame)
        d.clear(); d.reset_stats()
        d.render_frames(params, seeds); d.sync()
        st = d.stats()
        assert st.kernel_launches == 1 and st.launches == n
        sizes.append(st.wf_state_mib)
        assert_bit_equal(d.read_accum(), want, f"{n} frames in flight")
    assert sizes[0] == sizes[1] > 0 and sizes[0] % 3 == 0, sizes


def test_a_launch_alone_on_the_device_is_shaped_for_its_tail(gpu_device, monkeypatch):
    """Round 6 (glrtx.hip, launch_wgwf `shape`): a launch ends with a tail in which its last paths run out, and nothing overlaps that tail when the launch is alone on the
    device -- so a single frame issued to an idle device keeps fewer paths per workgroup alive (more, shorter helpings) than the same frame issued while the previous one is
    still rendering (one helping, handed over to the next launch), fewer still when a path item is several samples long -- and all of them render the same image."""
    d = gpu_device
    monkeypatch.setenv("GLRTX_NO_FEED", "1")  # (every call a launch of its own: the second of two back-to-back calls is an overlapped launch, not a fed one)
    scene, params = scenes.config_c2(width=1920, height=1080, max_depth=5, subdiv=1)
    seeds = _seeds(3)
    want, _ = gpu_render(d, scene, params, frames=seeds)
    for k in range(8):  # (every pipe slot has its buffers: an allocation between two calls below could outlast the first one's launch)
        d.render(dict(params, seed=seeds[0]))
    d.clear(); d.sync()
    d.render(dict(params, seed=seeds[0])); d.sync()
    lone = d.stats().wf_state_mib
    d.render(dict(params, seed=seeds[1])); d.render(dict(params, seed=seeds[2])); busy = d.stats().wf_state_mib; d.sync()
    assert_bit_equal(d.read_accum(), want, "lone and overlapped single-frame launches")
    assert 0 < lone < busy, (lone, busy)  # (512 against 2048 paths per workgroup on a 256-CU device)
    d.clear(); d.sync()
    d.render_frames(params, seeds[:2]); d.sync()   # (two frames in one plain launch: two helpings of 2048)
    assert lone < d.stats().wf_state_mib, (d.stats().wf_state_mib, lone)
    want2, _ = gpu_render(d, scene, params, frames=seeds[:2])
    d.clear(); d.render_frames(params, seeds[:2]); d.sync()
    assert_bit_equal(d.read_accum(), want2, "two frames in one short plain launch")
    ref = None
    from oracle import pt_oracle
    small_scene, small = scenes.config_c1(width=96, height=64, max_depth=4, n_samples=3, subdiv=1)
    for sd in seeds:
        ref, _ = pt_oracle.render(small_scene, dict(small, seed=sd), accum=ref)
    for bp in ("4096", "512", "256"):
        monkeypatch.setenv("GLRTX_BLOCK_PATHS", bp)
        got, _ = gpu_render(d, small_scene, small, frames=seeds)
        assert_bit_equal(got, ref, f"block_paths {bp}")


@pytest.mark.parametrize("w,h", [(1, 1), (3, 2), (17, 1), (1, 33), (9, 9)])
def test_tiny_images_single_and_in_flight(gpu_device, w, h):
    """Images far smaller than a tile / a workgroup's path set: one frame per launch and four in flight vs the oracle."""
    from oracle import pt_oracle
    d = gpu_device
    scene, params = scenes.config_c1(width=w, height=h, max_depth=4, n_samples=2, subdiv=1)
    seeds = _seeds(4)
    ref = None
    for sd in seeds:
        ref, _ = pt_oracle.render(scene, dict(params, seed=sd), accum=ref)
    seq, _ = gpu_render(d, scene, params, frames=seeds)
    assert_bit_equal(seq, ref, f"{w}x{h} consecutive launches")
    d.clear(); d.render_frames(params, seeds); d.sync()
    assert_bit_equal(d.read_accum(), ref, f"{w}x{h} frames in flight")
    # the path state is sized by the workgroups this launch has, not by the device (it was 768 MiB for a 1x1 image)
    assert 0 < d.stats().wf_state_mib <= 1, d.stats().wf_state_mib


def test_untraced_rays_are_a_subset_and_disappear_without_lights(gpu_device):
    """stats.rays counts the reference's intersect() executions (== oracle); rays_untraced are the shadow rays among them
    whose light test cannot change the radiance.  A scene without emitters has shadow rays (the reference still calls
    sampleDirect) whose contribution is always zero: all of them are untraced."""
    from oracle import pt_oracle
    d = gpu_device
    scene, params = scenes.config_c2(width=160, height=90, max_depth=4, subdiv=1)
    _, st = gpu_render(d, scene, params)
    _, ref_rays = pt_oracle.render(scene, params)
    assert st.rays == ref_rays and 0 < st.rays_untraced < st.rays // 2
    golden = load_golden("no_lights")
    acc, st = gpu_render(d, golden[0], golden[1], golden[3])
    assert st.rays_untraced > 0


def test_more_materials_than_fit_in_lds(gpu_device):
    """> 256 materials: the material table stays in global memory instead of being staged into LDS."""
    from oracle import pt_oracle
    rng = np.random.default_rng(77)
    pos, nrm, _ = scenes.random_triangles(600, 78, 1.0, 1.0)
    b = scenes.SceneBuilder()
    ids = [b.add_material(scenes.diffuse(tuple(rng.uniform(0.1, 0.9, 3)))) for _ in range(280)]
    ids += [b.add_material(scenes.conductor(scenes.COPPER["eta"], scenes.COPPER["kappa"], float(rng.uniform(0.05, 0.5)))) for _ in range(40)]
    ids += [b.add_material(scenes.emitter(tuple(rng.uniform(1.0, 9.0, 3)))) for _ in range(20)]
    b.add_mesh(pos, nrm, np.asarray(ids)[rng.integers(0, len(ids), 600)])
    scene = b.build("sah")
    assert scene["mat"].reshape(-1, 18).shape[0] == 340
    c2w, s2c = scenes.camera((1.0, 0.8, 3.0), (0, 0, 0), (0, 1, 0), 45.0, 72, 40)
    params = scenes.make_params(c2w, s2c, 72, 40, 6, 2)
    ref, ref_rays = pt_oracle.render(scene, params)
    d = gpu_device
    try:
        for variant in (2, 0, 1):
            d.set_variant(variant)
            acc, st = gpu_render(d, scene, params)
            assert st.rays == ref_rays
            assert_bit_equal(acc, ref, f"340 materials, variant {variant}")
    finally:
        d.set_variant(2)


# ---------------------------------------------------------------- groups (multi-GPU through the C ABI)
@pytest.mark.parametrize("n_members", [1, 2, 3, 8])
def test_group_renders_the_single_context_image(gpu_device, n_members):
    """glrtx_group with n members (all on this GPU: the partition arithmetic, the concurrent streams, the stripe gather and the
    full-frame resolve are what is under test) == one context, bit for bit: accumulator, ray count and RGBA8 bytes; with single
    launches and with frames in flight; at a height that leaves ragged stripes."""
    scene, params = scenes.config_c1(160, 104, max_depth=4, n_samples=1, subdiv=1)  # 104 rows = 6.5 stripes
    seeds = [host.frame_seed(f) for f in range(5)]
    d = gpu_device
    ref, ref_st = gpu_render(d, scene, params, frames=seeds)
    ref_rays = int(ref_st.rays)
    ref8 = d.resolve_rgba8(2.2, True)
    g = device.Group([0] * n_members)
    try:
        assert g.size() == n_members
        g.upload_scene(scene)
        g.resize(160, 104)
        g.member_call(g.L.glrtx_count_rays, 1)
        g.render(dict(params, seed=seeds[0]))
        g.render_frames(params, seeds[1:4])
        g.render(dict(params, seed=seeds[4]))
        g.sync()
        st = g.stats()
        assert st.owned_rows == 104 and st.rays == ref_rays
        assert_bit_equal(g.read_accum(), ref, f"group of {n_members}")
        assert g.gather_copies() <= min(2 * n_members, 13)  # one strided copy per member (+ one for a partial last stripe), never one per stripe
        assert np.array_equal(g.resolve_rgba8(2.2, True), ref8)
        g.clear()
        g.render(dict(params, seed=seeds[0]))
        one, _ = gpu_render(d, scene, params)
        assert_bit_equal(g.read_accum(), one, "after clear")
    finally:
        g.close()


@pytest.mark.parametrize("w,h,n_members", [(50, 38, 3), (33, 7, 2), (64, 8, 8), (40, 9, 8)])
def test_group_with_a_partial_last_stripe(gpu_device, w, h, n_members):
    """Heights that are not a multiple of the 8-row stripe: the last stripe is short, some members own nothing at all."""
    scene, params = scenes.config_c1(w, h, max_depth=3, n_samples=2, subdiv=1)
    seeds = [host.frame_seed(f) for f in range(3)]
    ref, ref_st = gpu_render(gpu_device, scene, params, frames=seeds)
    ref8 = gpu_device.resolve_rgba8(2.2, True)
    g = device.Group([0] * n_members)
    try:
        g.upload_scene(scene); g.resize(w, h)
        g.member_call(g.L.glrtx_count_rays, 1)
        g.render_frames(params, seeds)
        assert_bit_equal(g.read_accum(), ref, f"{w}x{h} over {n_members} members")
        assert g.stats().rays == ref_st.rays and g.stats().owned_rows == h
        assert np.array_equal(g.resolve_rgba8(2.2, True), ref8)
        assert g.gather_copies() <= 2 * n_members
    finally:
        g.close()


@pytest.mark.parametrize("cfg,kw,frames", [("headline", {}, 2), ("c4", dict(n_samples=16), 1)], ids=["1080p", "4k_16spp"])
def test_group_of_eight_at_full_size(gpu_device, cfg, kw, frames):
    """BASELINE's multi-GPU shapes through the C-ABI group (eight members, here all on one GPU): 1920x1080 / 8 bounces and config 4,
    3840x2160 / 8 bounces / 16 spp -- the gathered frame equals the single-context frame bit for bit, with 8 gather copies (1080 rows =
    135 stripes, 2160 = 270: round 2 issued one copy per stripe)."""
    scene, params = scenes.CONFIGS[cfg](**kw)
    seeds = [host.frame_seed(f) for f in range(frames)]
    ref, _ = gpu_render(gpu_device, scene, params, frames=seeds, count_rays=False)
    g = device.Group([0] * 8)
    try:
        g.upload_scene(scene); g.resize(params["width"], params["height"])
        if frames > 1:
            g.render_frames(params, seeds)
        else:
            g.render(dict(params, seed=seeds[0]))
        assert_bit_equal(g.read_accum(), ref, f"{cfg} over 8 members")
        assert g.gather_copies() == 8
    finally:
        g.close()


def test_group_error_paths():
    with pytest.raises(device.GlrtxError):
        device.Group([0, 99])  # no such device
    g = device.Group([0, 0])
    try:
        with pytest.raises(device.GlrtxError) as e:
            g.render(scenes.config_c1(32, 32, subdiv=1)[1])
        assert e.value.code == device.GLRTX_EINVAL and "context 0" in str(e.value) and "no scene" in str(e.value)
    finally:
        g.close()


def test_leaving_the_wavefront_kernel_is_visible_in_the_stats(gpu_device):
    """u_maxDepth > 255 (and extension scenes) run on the persistent megakernel: stats.variant_last / fallback_last / fallback_launches say so."""
    d = gpu_device
    scene, params = scenes.config_c1(48, 32, max_depth=4, subdiv=1)
    _, st = gpu_render(d, scene, params)
    assert st.variant_last == 2 and st.fallback_last == 0 and st.fallback_launches == 0
    _, st = gpu_render(d, scene, dict(params, max_depth=256), frames=[host.frame_seed(0), host.frame_seed(1)])
    assert st.variant_last == 1 and st.fallback_last == device.FALLBACK_DEPTH and st.fallback_launches == 2
    d.set_extensions(device.EXT_WHITTED)
    try:
        _, st = gpu_render(d, scene, params)
        assert st.variant_last == 1 and st.fallback_last == device.FALLBACK_EXTENSIONS and st.fallback_launches == 1
    finally:
        d.set_extensions(0)
    d.set_variant(1)
    try:
        _, st = gpu_render(d, scene, params)
        assert st.variant_last == 1 and st.fallback_last == 0 and st.fallback_launches == 0  # asked for, not a fallback
    finally:
        d.set_variant(2)


def test_render_calls_return_before_the_device_is_done(gpu_device, monkeypatch):
    """glrtx_render is asynchronous (include/glrtx.h): consecutive calls are enqueued without waiting for the previous launch -- the host
    returns from several 1080p launches in a fraction of the time the device needs for them -- and the per-launch kernel times are
    folded into the stats afterwards (ring of event triples, also beyond its 16 entries).  (One launch per call: GLRTX_NO_FEED=1; with fed launches -- the
    default on the context's own stream, tests/test_gpu_feed.py -- the calls of such a burst are not even launches.)"""
    import time
    monkeypatch.setenv("GLRTX_NO_FEED", "1")
    d = gpu_device
    scene, params = scenes.config_headline()
    d.upload_scene(scene); d.set_partition(0, 1, 16); d.resize(params["width"], params["height"]); d.count_rays(False)
    d.render(dict(params, seed=host.frame_seed(0))); d.sync(); d.reset_stats()
    n = 6
    t0 = time.perf_counter()
    for f in range(n):
        d.render(dict(params, seed=host.frame_seed(1 + f)))
    t_enqueue = time.perf_counter() - t0
    d.sync()
    t_all = time.perf_counter() - t0
    st = d.stats()
    assert st.kernel_launches == n and st.launches == n
    device_s = st.kernel_ms_total * 1e-3
    assert device_s > 0.004                      # six 1080p frames: > 4 ms of device time
    assert t_enqueue < 0.5 * device_s, (t_enqueue, device_s, t_all)   # the host did not wait for the launches it issued
    d.reset_stats()
    for f in range(40):                          # more launches in flight than the ring holds: the oldest are folded on the way
        d.render(dict(params, seed=host.frame_seed(f)))
    d.sync()
    st = d.stats()
    assert st.kernel_launches == 40 and st.kernel_ms_total > 0.02


def test_overlapped_launches_next_to_a_callers_own_streams(gpu_device, monkeypatch):
    """One launch per frame (the reference's cadence) while the caller keeps two more streams of its own busy with small kernels between the render calls: the
    image is the quiet run's bit for bit, consecutive launches still overlap on the device (glrtx_stats.pipe_resident_max: how many internal slots had a render
    kernel running at once), and the time per frame stays within 15 % of the quiet run's.  Then the slots are squeezed by the memory budget: fewer slots,
    none at all (un-piped launches on the context's stream) -- the same image every time, never a failed render."""
    import time
    import torch
    monkeypatch.setenv("GLRTX_NO_FEED", "1")  # (the overlapped single-frame launches themselves: what a render-resolve-save loop and a caller's stream get)
    d = gpu_device
    scene, params = scenes.config_headline()
    d.upload_scene(scene); d.set_partition(0, 1, 16); d.resize(params["width"], params["height"]); d.count_rays(False)

    caller_streams = [torch.cuda.Stream() for _ in range(2)]
    caller_data = [torch.zeros(1 << 16, device="cuda") for _ in caller_streams]

    def run(n, busy):
        d.clear(); d.reset_stats()
        streams, xs = (caller_streams, caller_data) if busy else ([], [])
        d.sync(); torch.cuda.synchronize()
        t = time.perf_counter()
        for f in range(n):
            d.render(dict(params, seed=host.frame_seed(f)))
            for st_, x in zip(streams, xs):
                with torch.cuda.stream(st_):
                    x.add_(1.0)
        d.sync(); torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3, d.read_accum(), d.stats()

    run(8, False)  # every slot has its buffers
    run(8, True)   # ... and the caller's kernel is loaded (loading a code object stalls the device once: not what is measured)
    ms_q, img_q, st_q = run(40, False)
    ms_b, img_b, st_b = run(40, True)
    assert_bit_equal(img_b, img_q, "busy caller streams")
    assert st_q.pipe_slots == 6 and st_q.pipe_resident_max >= 2, (st_q.pipe_slots, st_q.pipe_resident_max)
    assert st_b.pipe_resident_max >= 2, st_b.pipe_resident_max
    assert ms_b <= 1.15 * ms_q, (ms_b, ms_q)
    # ~0.8 GB per slot at 1080p: a 2 GiB budget leaves two slots, 512 MiB none
    monkeypatch.setenv("GLRTX_FRAMES_BUDGET_MB", "2048")
    _, img2, st2 = run(12, False)
    assert_bit_equal(img2, run(12, False)[1], "two slots, repeated")
    assert st2.pipe_slots == 2 and st2.pipe_resident_max <= 2
    monkeypatch.setenv("GLRTX_FRAMES_BUDGET_MB", "512")
    _, img0, st0 = run(12, False)
    assert st0.pipe_slots == 0 and st0.pipe_resident_max == 0 and st0.kernel_launches == 12
    monkeypatch.delenv("GLRTX_FRAMES_BUDGET_MB")
    _, img6, st6 = run(12, False)
    assert st6.pipe_slots == 6
    assert_bit_equal(img0, img6, "un-piped launches")
    assert_bit_equal(img2, img6, "two slots")


@BOTH_INSTANTIATIONS
@pytest.mark.parametrize("cfg,kw", [("c2", dict(width=320, height=200, max_depth=8, n_samples=1)), ("c2", dict(width=97, height=61, max_depth=5, n_samples=3)),
                                    ("c5", dict(width=256, height=144, max_depth=4, n_samples=1, n=4000))])
@pytest.mark.parametrize("fed", [False, True], ids=["overlapped", "fed"])
def test_many_overlapped_single_frame_launches_equal_one_launch_of_all_frames(gpu_device, monkeypatch, cfg, kw, count_rays, fed):
    """26 back-to-back glrtx_render calls -- the internal slots (stream, path state, queues, planes, tile counter each) come round four times,
    every launch with the full grid and without guided self-scheduling, workgroups of up to six frames resident side by side -- against the
    same 26 frames as one glrtx_render_frames launch (itself pinned to the oracle above): accumulator and ray count, bitwise.  Both as 26 overlapped
    launches (GLRTX_NO_FEED=1: round 3's form, what a caller's stream still gets) and as the default of round 6, where the calls behind the first feed a running launch."""
    if not fed:
        monkeypatch.setenv("GLRTX_NO_FEED", "1")
    d = gpu_device
    scene, params = scenes.CONFIGS[cfg](**kw)
    seeds = _seeds(26, start=3)
    seq, st_seq = gpu_render(d, scene, params, frames=seeds, count_rays=count_rays)
    assert st_seq.launches == 26 and (st_seq.kernel_launches == 26 if not fed else 1 <= st_seq.kernel_launches <= 26)
    d.clear(); d.reset_stats()
    d.render_frames(params, seeds); d.sync()
    assert_bit_equal(seq, d.read_accum(), f"{cfg} {kw}: 26 overlapped launches vs one launch of 26 frames")
    assert d.stats().rays == st_seq.rays and (st_seq.rays > 0) == count_rays


def _closed_box(width, height, max_depth, n_samples):
    """A closed room of albedo-0.98 walls with a small lamp: paths end almost only through Russian roulette (survival 0.95 per
    bounce beyond depth 2), so path lengths have a long tail."""
    b = scenes.SceneBuilder()
    wall = b.add_material(scenes.diffuse((0.98, 0.98, 0.98)))
    lamp = b.add_material(scenes.emitter((20.0, 20.0, 20.0)))
    lo, hi = -2.0, 2.0
    b.add_mesh(*scenes.quad((lo, lo, hi), (hi - lo, 0, 0), (0, 0, lo - hi)), wall)   # floor, normal +y
    b.add_mesh(*scenes.quad((lo, hi, lo), (hi - lo, 0, 0), (0, 0, hi - lo)), wall)   # ceiling, normal -y
    b.add_mesh(*scenes.quad((lo, lo, lo), (hi - lo, 0, 0), (0, hi - lo, 0)), wall)   # back, normal +z
    b.add_mesh(*scenes.quad((hi, lo, hi), (lo - hi, 0, 0), (0, hi - lo, 0)), wall)   # front, normal -z
    b.add_mesh(*scenes.quad((lo, lo, hi), (0, 0, lo - hi), (0, hi - lo, 0)), wall)   # left, normal +x
    b.add_mesh(*scenes.quad((hi, lo, lo), (0, 0, hi - lo), (0, hi - lo, 0)), wall)   # right, normal -x
    b.add_mesh(*scenes.quad((-0.5, hi - 0.01, -0.5), (1, 0, 0), (0, 0, 1)), lamp)                # lamp under the ceiling, facing down
    sc = b.build()
    c2w, s2c = scenes.camera((0.0, 0.0, 1.8), (0, 0, 0), (0, 1, 0), 70.0, width, height)
    return sc, scenes.make_params(c2w, s2c, width, height, max_depth, n_samples)


@pytest.mark.parametrize("max_depth", [255, 256, 300])
def test_depth_beyond_the_packed_path_state_matches_the_oracle(gpu_device, max_depth):
    """The wavefront kernel keeps depth in 8 bits of its packed path state; u_maxDepth > 255 is rendered by the persistent
    megakernel instead (glrtx_render).  Both sides of that switch against the oracle, on a scene with long paths, single
    launches and glrtx_render_frames."""
    from oracle import pt_oracle
    sc, pr = _closed_box(48, 48, max_depth, 4)
    ref, ref_rays = pt_oracle.render(sc, pr)
    acc, st = gpu_render(gpu_device, sc, pr)
    assert int(st.rays) == ref_rays and ref_rays > 20 * 48 * 48 * 4  # long paths indeed
    assert_bit_equal(acc, ref, f"depth {max_depth}")
    d = gpu_device
    d.clear(); d.reset_stats()
    d.render_frames(pr, [pr["seed"], host.frame_seed(3)])
    d.sync()
    pt_oracle.render(sc, dict(pr, seed=host.frame_seed(3)), accum=ref)
    assert_bit_equal(d.read_accum(), ref, f"depth {max_depth}, two frames in one call")


def test_short_quotients_equal_the_ieee_quotient_on_every_float():
    """pt_kernel.hip.h computes 1 / det of the triangle tests as v_rcp_f32 + one Newton step (rcp_newton), every other 1 / x with the raw v_rcp_f32 result
    for zeros, denormals and infinities (frcp), and x / PI as a multiplication corrected by one residual step (div_pi).  tools/ubench/rcp_exact.hip (built
    by __graft_entry__.build() with the kernels' own float mode: fp32 denormals flushed, as the reference's GL implementation runs) compares each of them
    with the compiler's correctly rounded division for EVERY float bit pattern on the device."""
    import re
    import subprocess
    exe = PKG / "lib" / "rcp_exact"
    assert exe.exists(), f"{exe} not built: run __graft_entry__.build()"
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert re.search(r"^patterns 4294967296$", r.stdout, re.M), r.stdout
    m = re.search(r"newton: (\d+) mismatches among normal finite x \((\d+) of them above 2\^126\), (\d+) among zeros, denormals, infinities and NaNs", r.stdout)
    assert m, r.stdout
    assert int(m.group(1)) == 0 and int(m.group(2)) == 0, "rcp_newton is the IEEE quotient of every normal finite float"
    assert int(m.group(3)) == (1 << 24) + 2, "... and of nothing else but NaNs: 2^24 zeros and denormals, two infinities (its callers reject |det| < EPS first)"
    assert re.search(r"^frcp: 0 mismatches$", r.stdout, re.M), r.stdout
    assert re.search(r"^div_pi: 0 mismatches$", r.stdout, re.M), r.stdout
    # ... and the short form by itself on every pattern of its range, the inclusive ends 2^-100 and 2^120 among them (the wave-level branch of div_pi hides those)
    m = re.search(r"^div_pi short form alone: 0 mismatches among the (\d+) patterns of its range", r.stdout, re.M)
    assert m and int(m.group(1)) == 2 * (0x7B800000 - 0x0D800000 + 1) + 1, r.stdout  # +0 and both signs of 2^-100 .. 2^120 (-0 takes the full division: the residual step would make it +0)


def test_contexts_driven_from_concurrent_host_threads(gpu_device):
    """include/glrtx.h: one host thread drives a context -- and several contexts, each with its thread, share the device: three threads upload different scenes
    (tree, list scan, extension spheres) into contexts of their own and render frames at the same time; every image is the oracle's."""
    import threading
    from oracle import pt_oracle
    jobs = [("c1", scenes.config_c1(96, 64, max_depth=4, n_samples=1, subdiv=1), None),
            ("c3 chain", scenes.config_c3(64, 48, max_depth=2, n=300, bvh="chain"), None),
            ("spheres", scenes.config_spheres(64, 48, max_depth=4, n_samples=1)[:2], scenes.config_spheres(64, 48, max_depth=4, n_samples=1)[2])]
    seeds = [host.frame_seed(f) for f in range(6)]
    refs = []
    for name, (sc, pr), sph in jobs:
        ref = None
        for sd in seeds:
            ref, _ = pt_oracle.render(sc, dict(pr, seed=sd), accum=ref, spheres=sph)
        refs.append(ref)
    results, errors = [None] * len(jobs), []
    start = threading.Barrier(len(jobs))

    def work(i):
        try:
            name, (sc, pr), sph = jobs[i]
            d = device.Device()
            try:
                d.upload_scene(sc)
                if sph is not None:
                    d.upload_spheres(sph)
                d.resize(pr["width"], pr["height"])
                start.wait(timeout=60)
                for rep in range(3):  # the same six frames three times over: launches of the three contexts interleave on the device
                    d.clear()
                    for k, sd in enumerate(seeds):
                        if k % 2:
                            d.render(dict(pr, seed=sd))
                        else:
                            d.render_frames(pr, [sd])
                    d.sync()
                    acc = d.read_accum()
                    if results[i] is None:
                        results[i] = acc
                    elif not np.array_equal(results[i].view(np.uint32), acc.view(np.uint32)):
                        errors.append(f"{name}: repetition {rep} differs from the first")
            finally:
                d.close() if hasattr(d, "close") else None
        except Exception as e:  # noqa: BLE001
            errors.append(f"{jobs[i][0]}: {type(e).__name__}: {e}")

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for (name, _, _), got, ref in zip(jobs, results, refs):
        assert_bit_equal(got, ref, f"{name}, rendered next to two other contexts")


def test_hit_histogram_and_the_measured_child_order(gpu_device):
    """glrtx_hit_histogram counts, per triangle, the closest hits of the path rays of one frame -- their sum is the oracle's count of path rays that hit something, the
    context's accumulator and statistics are untouched -- and a tree whose forks are re-ordered by those counts (glrt_bvh_order_by_hits, with the shadow rays' share)
    renders what the oracle renders for that same tree, bit for bit."""
    from oracle import pt_oracle
    d = gpu_device
    scene, params = scenes.CONFIGS["c5"](width=192, height=108, n=3000)
    acc0, st0 = gpu_render(d, scene, params)
    hist = d.hit_histogram(params, scene["tri"].shape[0])
    st1 = d.stats()
    assert st1.launches == st0.launches and st1.kernel_launches == st0.kernel_launches and st1.rays == st0.rays
    assert_bit_equal(d.read_accum(), acc0, "the accumulator after a calibration frame")
    # every shaded hit of the frame is counted once: rays of the oracle = path rays + shadow rays; hits <= path rays
    _, ref_rays = pt_oracle.render(scene, params)
    assert 100 < int(hist.sum()) <= ref_rays  # (a sparse soup of 3000 triangles: most rays of this frame miss)
    nodes, exchanged = host.order_by_hits(scene["bvh"], hist, scene["tri"], scene["mat"])
    assert exchanged > 0
    sc2 = dict(scene, bvh=nodes)
    ref, ref_rays2 = pt_oracle.render(sc2, params)
    acc, st = gpu_render(d, sc2, params)
    assert st.rays == ref_rays2 == ref_rays
    assert_bit_equal(acc, ref, "tree re-ordered by measured hits")
