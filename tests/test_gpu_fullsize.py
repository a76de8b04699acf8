"""Parity at BASELINE.json's FULL sizes: the HIP path (through the C ABI) against the ORACLE -- not against itself.

tests/test_gpu_parity.py compares with the oracle up to 480x270 and checks the full sizes through properties of the device path
alone.  What only exists at full size -- 2 M (1080p) to 8 M (4K) path ids per frame times the frames in flight, the queue slices of
every workgroup, the chunking of a launch by the memory budget, the plane offsets of 4K x 16 spp -- is compared here with the
reference's arithmetic (oracle/pt_oracle.c, pinned by the llvmpipe fixtures; raytrace.frag:565-614 is what one pixel does):

  * whole frames, bit for bit and with the ray count: the headline (1920x1080, 8 bounces), config 2 (depth 4), config 5 (100 k
    triangles, the CPU SAH tree and the two trees the GPU builds); the headline also as eight overlapped single-frame launches;
  * row bands (first, middle and last stripes) where the oracle needs minutes for a whole frame: config 3 at 1080p (10 k triangles
    as a chain = brute force) and config 4 at 3840x2160 with 16 spp in one pass and as 16 accumulated frames;
  * every one of them as a single launch (both compilations of the kernel), as 8 (16) frames in flight, as an 8-rank partition
    stitched together, and through the 8-member C-ABI group.
The oracle renders a 1080p headline frame in ~0.35 s on the GPU box's 16 cores; the module costs about two minutes there."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from glrt_amd import device, dist, host, scenes

pytestmark = pytest.mark.gpu

N_IN_FLIGHT = 8


def _seeds(n):
    return [host.frame_seed(f) for f in range(n)]


def _oracle(scene, params, seeds, rows=None):
    """The oracle run frame after frame into one accumulator (the reference's accumulation loop, window.cpp:214-252): (accum, rays).
    rows = (y0, y1): only that band is rendered (and only it is meaningful in the result)."""
    from oracle import pt_oracle
    acc, rays = None, 0
    for sd in seeds:
        acc, n = pt_oracle.render(scene, dict(params, seed=sd), accum=acc, rows=rows)
        rays += n
    return acc, rays


def _single(d, scene, params, seed, count):
    d.upload_scene(scene)
    d.set_partition(0, 1, 8)
    d.resize(params["width"], params["height"])
    d.reset_stats()
    d.count_rays(count)
    d.render(dict(params, seed=seed))
    d.sync()
    return d.read_accum(), d.stats()


def _in_flight(d, scene, params, seeds, count=False):
    d.upload_scene(scene)
    d.set_partition(0, 1, 8)
    d.resize(params["width"], params["height"])
    d.reset_stats()
    d.count_rays(count)
    d.render_frames(params, seeds)
    d.sync()
    return d.read_accum(), d.stats()


def _partitioned(d, scene, params, seeds, world=8, stripe=8):
    """The image of `world` ranks, each rendering its interleaved stripes with the frames in flight, stitched on the host."""
    h = params["height"]
    out = np.zeros((h, params["width"], 4), np.float32)
    d.upload_scene(scene)
    try:
        for rank in range(world):
            d.set_partition(rank, world, stripe)
            d.resize(params["width"], h)
            d.count_rays(False)
            d.render_frames(params, seeds)
            d.sync()
            ys = dist.owned_rows(rank, world, stripe, h)
            assert np.array_equal(d.local_rows_y(), ys)
            out[ys] = d.read_accum()
    finally:
        d.set_partition(0, 1, 8)
    return out


def _group(scene, params, seeds, members=8):
    g = device.Group([0] * members)
    try:
        g.upload_scene(scene)
        g.resize(params["width"], params["height"])
        g.render_frames(params, seeds)
        img = g.read_accum()
        assert g.gather_copies() == members
        return img
    finally:
        g.close()


@pytest.mark.parametrize("cfg", ["headline", "c2", "c5"])
def test_whole_frames_at_full_size_equal_the_oracle(gpu_device, cfg):
    """1920x1080 whole frames: headline (8 bounces), config 2 (4 bounces), config 5 (100 k triangles)."""
    d = gpu_device
    scene, params = scenes.CONFIGS[cfg]()
    assert (params["width"], params["height"]) == (1920, 1080)
    seeds = _seeds(N_IN_FLIGHT)
    ref1, rays1 = _oracle(scene, params, seeds[:1])
    # one launch, one frame: the kernel with ray counting (the exact count of intersect() executions) and the one bench.py times
    acc, st = _single(d, scene, params, seeds[0], True)
    assert st.rays == rays1
    assert_bit_equal(acc, ref1, f"{cfg} 1080p, one frame, counting kernel")
    acc, st = _single(d, scene, params, seeds[0], False)
    assert_bit_equal(acc, ref1, f"{cfg} 1080p, one frame, timed kernel")
    # the accumulation loop: N frames, on the device as ONE launch with N frames in flight
    refn, raysn = _oracle(scene, params, seeds)
    acc, st = _in_flight(d, scene, params, seeds, count=True)
    assert st.rays == raysn and st.launches == N_IN_FLIGHT
    assert_bit_equal(acc, refn, f"{cfg} 1080p, {N_IN_FLIGHT} frames in flight, counting kernel")
    acc, _ = _in_flight(d, scene, params, seeds, count=False)
    assert_bit_equal(acc, refn, f"{cfg} 1080p, {N_IN_FLIGHT} frames in flight, timed kernel")
    assert np.all(acc[..., 3] == float(N_IN_FLIGHT))
    # the multi-GPU shapes: 8 ranks' interleaved 8-row stripes stitched, and the C-ABI group of 8
    assert_bit_equal(_partitioned(d, scene, params, seeds), refn, f"{cfg} 1080p, 8-rank partition, frames in flight")
    assert_bit_equal(_group(scene, params, seeds), refn, f"{cfg} 1080p, group of 8, frames in flight")


def test_overlapped_single_frame_launches_at_full_size_equal_the_oracle(gpu_device):
    """The reference's own cadence -- one glrtx_render call per frame (window.cpp:121-169) -- at the headline's full size: eight calls issued back to back overlap on the
    device (six internal streams with buffers of their own, the planes added to the accumulator in call order); the accumulated image is the oracle's, bit for bit."""
    d = gpu_device
    scene, params = scenes.config_headline()
    seeds = _seeds(N_IN_FLIGHT)
    ref, rays = _oracle(scene, params, seeds)
    d.upload_scene(scene)
    d.set_partition(0, 1, 8)
    d.resize(params["width"], params["height"])
    for count in (True, False):
        d.clear(); d.reset_stats(); d.count_rays(count)
        for sd in seeds:
            d.render(dict(params, seed=sd))
        d.sync()
        st = d.stats()
        assert_bit_equal(d.read_accum(), ref, f"headline 1080p, {N_IN_FLIGHT} overlapped single-frame launches, counting={count}")
        if count:
            assert st.rays == rays
        assert st.pipe_slots >= 1 and st.launches == N_IN_FLIGHT


def test_config5_with_the_gpu_built_tree_at_full_size(gpu_device):
    """BASELINE config 5 as it is named: 100 k triangles, LINEAR BVH (built on the device), 1920x1080, 4 bounces -- against the oracle walking
    the same tree, and against the oracle on the CPU SAH tree (random triangles: no exact ties, so the tree cannot show in the image)."""
    from oracle import pt_oracle
    d = gpu_device
    scene, params = scenes.config_c5()
    nodes, depth, _ = d.build_lbvh(scene["vert"], scene["tri"])
    lb = dict(scene, bvh=nodes, bvh_depth=depth, bvh_kind="lbvh")
    seeds = _seeds(2)
    ref, rays = _oracle(lb, params, seeds)
    acc, st = _in_flight(d, lb, params, seeds, count=True)
    assert st.rays == rays and st.node_fetch_last == 1
    assert_bit_equal(acc, ref, "c5 1080p, GPU-built LBVH, vs the oracle on that tree")
    ref_sah, _ = _oracle(scene, params, seeds)
    assert_bit_equal(acc, ref_sah, "c5 1080p, LBVH image vs the oracle's SAH-tree image")
    # the binned SAH built on the device (glrtx_build_bvh_sah, round 5): the oracle walking that tree, and the same image again
    nodes, depth, _ = d.build_bvh_sah(scene["vert"], scene["tri"])
    sl = dict(scene, bvh=nodes, bvh_depth=depth, bvh_kind="sahl")
    ref, rays = _oracle(sl, params, seeds)
    acc, st = _in_flight(d, sl, params, seeds, count=True)
    assert st.rays == rays
    assert_bit_equal(acc, ref, "c5 1080p, device-built SAH tree, vs the oracle on that tree")
    assert_bit_equal(acc, ref_sah, "c5 1080p, device-built SAH tree vs the oracle's CPU-SAH-tree image")


def _bands(h, rows):
    """First, middle and last `rows` rows (stripe-aligned)."""
    mid = (h // 2) // 8 * 8
    return [(0, rows), (mid, mid + rows), (h - rows, h)]


def _assert_bands(img, scene, params, seeds, bands, what):
    for y0, y1 in bands:
        ref, _ = _oracle(scene, params, seeds, rows=(y0, y1))
        assert_bit_equal(img[y0:y1], ref[y0:y1], f"{what}, rows {y0}..{y1}")


def test_config3_brute_force_at_1080p_bands_equal_the_oracle(gpu_device):
    """Config 3 at its full size: 10,000 triangles as a chain (the list scan, csrc/scan_asm.hip.h), 1920x1080, 1 bounce.  The oracle needs ~1 s of 16 cores per
    16 rows, so bands: 64 rows each at the bottom, the middle and the top of a single frame; one stripe each of the 8-frame accumulation."""
    d = gpu_device
    scene, params = scenes.config_c3()
    assert "chain" in scene["bvh_kind"] and (params["width"], params["height"]) == (1920, 1080)
    seeds = _seeds(N_IN_FLIGHT)
    for count in (True, False):
        acc, st = _single(d, scene, params, seeds[0], count)
        _assert_bands(acc, scene, params, seeds[:1], _bands(1080, 64), f"c3 1080p, one frame, counting={count}")
    acc, _ = _in_flight(d, scene, params, seeds)
    bands = _bands(1080, 8)
    _assert_bands(acc, scene, params, seeds, bands, f"c3 1080p, {N_IN_FLIGHT} frames in flight")
    part = _partitioned(d, scene, params, seeds)
    assert_bit_equal(part, acc, "c3 1080p: 8-rank partition vs one rank (whole frame)")
    for y0, y1 in bands:  # (the oracle's bands were compared with `acc` above; the partition equals it everywhere)
        assert_bit_equal(part[y0:y1], acc[y0:y1], "c3 bands")


def test_config4_4k_16spp_bands_equal_the_oracle(gpu_device):
    """Config 4 at its full size, 3840x2160, 8 bounces, 16 spp: in ONE pass (u_nSamples = 16: 132.7 M path ids in one launch, chunked by the memory budget,
    sixteen planes per frame) and as SIXTEEN accumulated frames of 1 spp in flight; three 64-row bands of each against the oracle; the 8-rank partition and
    the group of 8 against the same bands."""
    d = gpu_device
    scene, params = scenes.config_c4()
    assert (params["width"], params["height"], params["n_samples"], params["max_depth"]) == (3840, 2160, 16, 8)
    bands = _bands(2160, 64)
    seeds = _seeds(2)
    for count in (True, False):
        acc, st = _single(d, scene, params, seeds[0], count)
        assert np.all(acc[..., 3] == 16.0)
        if count:
            assert st.paths == 3840 * 2160 * 16
        _assert_bands(acc, scene, params, seeds[:1], bands, f"c4 4K, 16 spp in one pass, counting={count}")
    # two such frames in flight (32 planes), whole, partitioned and through the group
    acc, _ = _in_flight(d, scene, params, seeds)
    _assert_bands(acc, scene, params, seeds, bands, "c4 4K, 2 frames x 16 spp in flight")
    _assert_bands(_partitioned(d, scene, params, seeds), scene, params, seeds, bands[1:2], "c4 4K, 2 x 16 spp, 8-rank partition")
    assert_bit_equal(_group(scene, params, seeds), acc, "c4 4K, 2 x 16 spp, group of 8 vs one context")
    # the reference's own cadence: 16 frames of 1 spp each, accumulated (window.cpp:239 hard-codes u_nSamples = 1)
    p1 = dict(params, n_samples=1)
    seeds16 = _seeds(16)
    acc, _ = _in_flight(d, scene, p1, seeds16)
    assert np.all(acc[..., 3] == 16.0)
    _assert_bands(acc, scene, p1, seeds16, bands, "c4 4K, 16 accumulated frames of 1 spp in flight")
    assert_bit_equal(_partitioned(d, scene, p1, seeds16), acc, "c4 4K, 16 frames, 8-rank partition vs one rank")


def test_headline_two_multi_frame_calls_back_to_back_against_the_oracle(gpu_device):
    """Round 6: two glrtx_render_frames(8) calls issued back to back at 1920x1080 -- the second is appended to the first one's launch while it
    runs (a fed launch: pt_kernel.hip.h FeedHost / FeedDev) -- then eight single glrtx_render calls behind them, against the oracle's accumulation of the same 24
    frames, ray count included; and the same calls with GLRTX_NO_FEED=1 (one launch per call).  tests/test_gpu_feed.py has the other burst shapes."""
    import os
    scene, params = scenes.config_headline()
    seeds = _seeds(24)
    ref, ref_rays = _oracle(scene, params, seeds)
    d = gpu_device
    for fed in (True, False):
        if not fed:
            os.environ["GLRTX_NO_FEED"] = "1"
        try:
            d.upload_scene(scene); d.set_partition(0, 1, 8); d.resize(params["width"], params["height"]); d.reset_stats(); d.count_rays(True)
            d.render_frames(params, seeds[:8])
            d.render_frames(params, seeds[8:16])
            for sd in seeds[16:]:
                d.render(dict(params, seed=sd))
            d.sync()
            st = d.stats()
            assert st.rays == ref_rays and st.launches == 24
            assert (st.feed_appended >= 8 and st.kernel_launches <= 3) if fed else (st.feed_appended == 0 and st.kernel_launches == 10)
            assert_bit_equal(d.read_accum(), ref, f"two 8-frame calls and eight single ones back to back, fed={fed}")
        finally:
            os.environ.pop("GLRTX_NO_FEED", None)
    d.count_rays(False)
