"""Fed launches (round 6; pt_kernel.hip.h: FeedHost / FeedDev, glrtx.hip: feed_append): back-to-back glrtx_render / glrtx_render_frames calls with the same camera on the
context's own stream run as ONE persistent launch -- the host publishes the frames of the later calls in host-coherent memory and the running kernel takes them itself.
Only scheduling changes: every image here is compared bit for bit with the oracle or with the same frames rendered one launch at a time by a context that never feeds
(GLRTX_NO_FEED=1), ray counts included -- across bursts of every shape, calls that seal the open launch, camera changes, launches that fill up, and bursts that race the
kernel's own closing of the feed."""
import os
import time

import numpy as np
import pytest

from conftest import assert_bit_equal
from glrt_amd import device, host, scenes

pytestmark = pytest.mark.gpu


def _seeds(n, f0=0):
    return [host.frame_seed(f0 + i) for i in range(n)]


@pytest.fixture(scope="module")
def plain_device():
    """A context whose launches are never fed (the round-5 behaviour): the reference every burst is compared with."""
    d = device.Device()
    yield d
    d.close()


def _setup(d, scene, params):
    d.upload_scene(scene); d.set_partition(0, 1, 16); d.resize(params["width"], params["height"]); d.clear(); d.reset_stats()


def _plain(plain_device, scene, params, seeds, count=False):
    """The frames one launch at a time on the context that never feeds."""
    d = plain_device
    _setup(d, scene, params); d.count_rays(count)
    for sd in seeds:
        d.render(dict(params, seed=sd)); d.sync()  # (a sync behind every call: nothing is ever open when the next one comes)
    img, st = d.read_accum(), d.stats()
    assert st.feed_launches == 0 and st.feed_appended == 0
    d.count_rays(False)
    return img, int(st.rays)


def test_a_burst_of_single_frame_calls_is_one_launch_and_the_oracles_image(gpu_device, plain_device):
    """48 glrtx_render calls back to back at 960x540: the second call finds the first launch still running and starts a fed launch, the rest are appended to it
    (glrtx_stats.feed_appended) -- and the accumulator equals the oracle's accumulation of the same 48 frames, ray count included."""
    from oracle import pt_oracle
    scene, params = scenes.CONFIGS["headline"](width=960, height=540)
    seeds = _seeds(48)
    ref, ref_rays = None, 0
    for sd in seeds[:12]:
        ref, n = pt_oracle.render(scene, dict(params, seed=sd), accum=ref)
        ref_rays += n
    d = gpu_device
    _setup(d, scene, params); d.count_rays(True)
    for sd in seeds[:12]:
        d.render(dict(params, seed=sd))
    d.sync()
    st = d.stats()
    assert st.rays == ref_rays
    assert_bit_equal(d.read_accum(), ref, "burst of 12 against the oracle")
    assert st.feed_launches >= 1 and st.feed_appended >= 6, (st.feed_launches, st.feed_appended)   # (how many depends on timing; most of the burst is appended)
    assert st.launches == 12
    # the whole burst against the plain context
    want, want_rays = _plain(plain_device, scene, params, seeds, count=True)
    _setup(d, scene, params); d.count_rays(True)
    for sd in seeds:
        d.render(dict(params, seed=sd))
    d.sync()
    st = d.stats()
    assert st.rays == want_rays and st.launches == 48 and st.feed_appended >= 24
    assert_bit_equal(d.read_accum(), want, "burst of 48 against one launch per frame")
    d.count_rays(False)


def test_multi_frame_calls_back_to_back_are_appended(gpu_device, plain_device):
    """Three glrtx_render_frames calls of 16 frames, then one of 5 and single frames: everything behind the first call is appended to its launch."""
    scene, params = scenes.CONFIGS["headline"](width=640, height=360)
    seeds = _seeds(16 * 3 + 5 + 3)
    want, _ = _plain(plain_device, scene, params, seeds)
    d = gpu_device
    _setup(d, scene, params)
    for k in range(3):
        d.render_frames(params, seeds[16 * k:16 * k + 16])
    d.render_frames(params, seeds[48:53])
    for sd in seeds[53:]:
        d.render(dict(params, seed=sd))
    d.sync()
    st = d.stats()
    assert st.feed_launches >= 1 and st.feed_appended >= 16 and st.launches == len(seeds)
    assert_bit_equal(d.read_accum(), want, "multi-frame calls appended")


def test_calls_that_look_at_the_accumulator_seal_the_open_launch(gpu_device, plain_device):
    """read / resolve / clear / sync between the calls of a burst: what each of them sees is exactly the frames issued before it -- nothing that is rendered later may end up
    in a launch that is already queued in front of the observer."""
    scene, params = scenes.CONFIGS["headline"](width=480, height=270)
    seeds = _seeds(20)
    d, q = gpu_device, plain_device
    _setup(d, scene, params); _setup(q, scene, params)
    for sd in seeds[:5]:
        d.render(dict(params, seed=sd)); q.render(dict(params, seed=sd)); q.sync()
    a5 = d.read_accum()                       # (seals; waits)
    assert_bit_equal(a5, q.read_accum(), "after 5 frames")
    for sd in seeds[5:9]:
        d.render(dict(params, seed=sd)); q.render(dict(params, seed=sd)); q.sync()
    rgba = d.resolve_rgba8(2.2, True)         # (seals; the resolve pass is queued behind the launch as it stands)
    for sd in seeds[9:12]:
        d.render(dict(params, seed=sd))
    assert np.array_equal(rgba, q.resolve_rgba8(2.2, True)), "resolve after 9 frames saw something else"
    for sd in seeds[9:12]:
        q.render(dict(params, seed=sd)); q.sync()
    d.sync()
    assert_bit_equal(d.read_accum(), q.read_accum(), "after 12 frames")
    # clear in the middle of a burst: only what follows it remains
    for sd in seeds[12:16]:
        d.render(dict(params, seed=sd))
    d.clear()
    for sd in seeds[16:]:
        d.render(dict(params, seed=sd))
    d.sync()
    q.clear()
    for sd in seeds[16:]:
        q.render(dict(params, seed=sd)); q.sync()
    assert_bit_equal(d.read_accum(), q.read_accum(), "clear inside a burst")


def test_camera_and_sampling_changes_end_the_open_launch(gpu_device, plain_device):
    scene, params = scenes.CONFIGS["headline"](width=480, height=270)
    moved = dict(params, c2w=np.asarray(params["c2w"], np.float32).copy())
    moved["c2w"].reshape(-1)[12] += 0.25  # (whatever element: another camera)
    variants = [params, moved, dict(params, max_depth=3), dict(params, n_samples=2), dict(params, aperture=0.05, focal=4.0), params]
    d, q = gpu_device, plain_device
    _setup(d, scene, params); _setup(q, scene, params)
    f = 0
    for pv in variants:
        for _ in range(4):
            sd = host.frame_seed(f); f += 1
            q.render(dict(pv, seed=sd)); q.sync()
    prepared = [device.make_params(dict(pv, seed=host.frame_seed(4 * i + k))) for i, pv in enumerate(variants) for k in range(4)]
    for pp in prepared:  # (back to back: one C call each)
        d.render(pp)
    d.sync()
    st = d.stats()
    assert st.launches == 24 and st.feed_launches >= 3 and st.feed_appended >= 6, (st.feed_launches, st.feed_appended)
    assert_bit_equal(d.read_accum(), q.read_accum(), "six cameras / sampling settings in one stream of calls")


def test_a_launch_that_is_full_hands_over_to_the_next(gpu_device, plain_device, monkeypatch):
    """With a small frames-in-flight budget a fed launch holds few frames: the burst runs as a chain of fed launches, still bit-identical."""
    scene, params = scenes.CONFIGS["headline"](width=320, height=180)
    seeds = _seeds(40)
    want, _ = _plain(plain_device, scene, params, seeds)
    monkeypatch.setenv("GLRTX_FEED_CAP", "5")  # (what a small frames-in-flight budget does: five frames per fed launch)
    d = gpu_device
    _setup(d, scene, params)
    for sd in seeds[:20]:
        d.render(dict(params, seed=sd))
    d.render_frames(params, seeds[20:])  # 20 frames in one call: cut into launches that fit
    d.sync()
    st = d.stats()
    assert st.feed_launches >= 4 and st.launches == 40
    assert_bit_equal(d.read_accum(), want, "chain of full launches")


def test_bursts_that_race_the_kernels_own_close(gpu_device, plain_device):
    """Small frames and pauses of the launch's own length between the calls: the kernel runs dry and closes its feed while the host is about to publish the next frame --
    the compare-and-swap decides, the frame goes into the old launch or into a new one, and either way the image is the same.  300 frames in bursts with pauses of
    0-400 us; the counters show that both outcomes occurred."""
    scene, params = scenes.CONFIGS["c1"](width=128, height=128, max_depth=4)
    n = 300
    seeds = _seeds(n)
    want, want_rays = _plain(plain_device, scene, params, seeds, count=True)
    d = gpu_device
    _setup(d, scene, params); d.count_rays(True)
    rng = np.random.default_rng(7)
    for i, sd in enumerate(seeds):
        d.render(dict(params, seed=sd))
        pause = float(rng.integers(0, 5)) * 100e-6
        t = time.perf_counter()
        while time.perf_counter() - t < pause:
            pass
    d.sync()
    st = d.stats()
    assert st.rays == want_rays and st.launches == n
    assert st.feed_appended >= 10 and st.kernel_launches >= 10, (st.feed_appended, st.kernel_launches)  # appended frames AND launches that had closed in time
    assert_bit_equal(d.read_accum(), want, "racing the close")
    d.count_rays(False)


def test_full_size_bursts_against_the_oracle(gpu_device):
    """1920x1080, the headline: 8 glrtx_render calls back to back, then two overlapped glrtx_render_frames(16) -- 40 frames -- against the oracle (VERDICT round 5, item 1:
    'two overlapped multi-frame launches against the oracle')."""
    from oracle import pt_oracle
    scene, params = scenes.CONFIGS["headline"]()
    seeds = _seeds(40, 1000)
    ref, ref_rays = None, 0
    for sd in seeds:
        ref, k = pt_oracle.render(scene, dict(params, seed=sd), accum=ref)
        ref_rays += k
    d = gpu_device
    _setup(d, scene, params); d.count_rays(True)
    for sd in seeds[:8]:
        d.render(dict(params, seed=sd))
    d.render_frames(params, seeds[8:24])
    d.render_frames(params, seeds[24:])
    d.sync()
    st = d.stats()
    assert st.rays == ref_rays and st.feed_appended >= 32
    assert_bit_equal(d.read_accum(), ref, "full-size bursts against the oracle")
    d.count_rays(False)


def test_a_callers_stream_is_never_fed(gpu_device, plain_device):
    """On a caller's stream the order of the caller's own work is the caller's: a frame rendered later must not end up in a launch queued in front of work the caller has
    enqueued since.  Launches there are never fed."""
    import torch
    scene, params = scenes.CONFIGS["headline"](width=320, height=180)
    d = gpu_device
    _setup(d, scene, params)
    s = torch.cuda.Stream()
    d.set_stream(s.cuda_stream)
    try:
        for sd in _seeds(6):
            d.render(dict(params, seed=sd))
        d.render_frames(params, _seeds(8, 6))
        d.sync()
        st = d.stats()
        assert st.feed_launches == 0 and st.feed_appended == 0
    finally:
        d.set_stream(0)
    want, _ = _plain(plain_device, scene, params, _seeds(14))
    assert_bit_equal(d.read_accum(), want, "caller's stream")


@pytest.mark.parametrize("n_frames", [64, 70])
def test_plain_launches_keep_their_seeds_in_lds_or_in_memory(gpu_device, plain_device, monkeypatch, n_frames):
    """An unfed multi-frame launch keeps the seeds of up to 64 frames in LDS (WfArgs::seeds_in_lds), a longer one reads them from memory: both give the image of
    one launch per frame."""
    monkeypatch.setenv("GLRTX_NO_FEED", "1")
    scene, params = scenes.CONFIGS["headline"](width=160, height=90)
    seeds = _seeds(n_frames, 500)
    want, _ = _plain(plain_device, scene, params, seeds)
    d = gpu_device
    _setup(d, scene, params)
    d.render_frames(params, seeds); d.sync()
    st = d.stats()
    assert st.kernel_launches == 1 and st.feed_launches == 0
    assert_bit_equal(d.read_accum(), want, f"{n_frames} frames in one plain launch")


def test_light_triangles_in_lds_or_in_memory(gpu_device, monkeypatch):
    """Up to 64 light triangles are staged into LDS (DevScene::lights_in_lds); GLRTX_NO_LDS_LIGHTS=1 (read at upload) and scenes with more lights read them from memory.
    Same image, bit for bit, against the oracle."""
    from oracle import pt_oracle
    scene, params = scenes.CONFIGS["headline"](width=160, height=90)
    ref, ref_rays = pt_oracle.render(scene, params)
    d = gpu_device
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("GLRTX_NO_LDS_LIGHTS", env)
        _setup(d, scene, params); d.count_rays(True)
        d.render(params); d.sync()
        assert d.stats().rays == ref_rays
        assert_bit_equal(d.read_accum(), ref, f"lights in LDS off={env}")
    monkeypatch.delenv("GLRTX_NO_LDS_LIGHTS")
    d.count_rays(False)
    # many lights (config 5's soup: every tenth triangle emits): more than fit, the global path
    scene5, params5 = scenes.CONFIGS["c5"](width=160, height=90, n=4000)
    assert scene5["light"].shape[0] > 64
    ref5, _ = pt_oracle.render(scene5, params5)
    _setup(d, scene5, params5)
    d.render(params5); d.sync()
    assert_bit_equal(d.read_accum(), ref5, "more lights than LDS holds")


def test_a_burst_behind_a_plain_launch_is_fed(gpu_device, plain_device):
    """A frame whose sample planes exceed the overlapped form's 1-GiB limit (1080p at 33 spp: 1.09 GB) is a plain launch on the context's stream.  Since the end of round 6 such a
    launch counts as the start of a burst as well (its completion event is the context's path state's): the calls behind it open a fed launch and are appended to it.
    (Config 4, one glrtx_render per frame: 51.5 -> 30.9 ms per frame, profiles/r06_launch_shapes.txt.)  Same image as one launch per frame."""
    scene, params = scenes.config_c2(width=1920, height=1080, max_depth=1, n_samples=33, subdiv=1)
    seeds = _seeds(4)
    want, _ = _plain(plain_device, scene, params, seeds)
    d = gpu_device
    _setup(d, scene, params)
    for sd in seeds:
        d.render(dict(params, seed=sd))
    d.sync()
    st = d.stats()
    assert st.launches == 4 and st.feed_launches >= 1 and st.kernel_launches < 4, (st.launches, st.feed_launches, st.kernel_launches, st.feed_appended)
    assert_bit_equal(d.read_accum(), want, "burst behind a plain launch")
