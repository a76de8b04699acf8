"""The CPU restatement (oracle/pt_oracle.c) against golden images produced by the reference's own
unmodified shader on Mesa llvmpipe (tests/golden/make_golden.py).  Bar: bit-exact float32."""
import numpy as np
import pytest

from conftest import assert_bit_equal, golden_names, load_golden
from oracle import pt_oracle


def oracle_render_fixture(scene, params, rows, frames):
    h, w = params["height"], params["width"]
    acc = np.zeros((h, w, 4), np.float32)
    seeds = frames if frames else [params["seed"]]
    for sd in seeds:  # read-modify-write accumulation replaces the reference's ping-pong FBOs
        p = dict(params, seed=sd)
        pt_oracle.render(scene, p, accum=acc, rows=rows)
    return acc[rows[0]:rows[1]]


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_golden(name):
    scene, params, rows, frames, rgb, cnt = load_golden(name)
    acc = oracle_render_fixture(scene, params, rows, frames)
    assert_bit_equal(acc[..., :3], rgb, f"{name} rgb")
    assert_bit_equal(acc[..., 3], cnt, f"{name} count")


def test_goldens_cover_survey_list():
    names = set(golden_names())
    for required in ("emitter_view", "f6_floor_wall", "conductor_quads", "c1_dof", "tris1500_chain", "tris1500_sah",
                     "c1_npot_50x38", "many_lights", "c1_pingpong3", "c1_zero_dir_rows", "no_lights", "media_front"):
        assert required in names
    for d in (1, 4, 8, 16):
        for s in (1, 16):
            assert f"c1_d{d}_spp{s}" in names


def test_chain_and_sah_bvh_give_identical_images():
    a = load_golden("tris1500_chain")
    b = load_golden("tris1500_sah")
    assert_bit_equal(a[4], b[4], "reference images, chain vs SAH")


def test_emitter_view_is_exact_emission():
    scene, params, rows, frames, rgb, cnt = load_golden("emitter_view")
    lit = rgb.sum(-1) > 0
    assert lit.sum() > 100
    assert np.all(rgb[lit] == np.array([3.0, 2.0, 1.0], np.float32))
    assert np.all(cnt == 1.0)


def test_f6_parallel_receiver_dark_perpendicular_lit():
    """with the lamp facing down, the floor (parallel to it) fails the NEE acceptance test
    almost everywhere, while the back wall (perpendicular) passes."""
    scene, params, rows, frames, rgb, cnt = load_golden("f6_floor_wall")
    acc = oracle_render_fixture(scene, dict(params, max_depth=1, n_samples=4), rows, [])
    lum = acc[..., :3].sum(-1)
    floor = lum[2:15, 8:56]    # bottom rows: floor
    wall = lum[20:42, 8:56]    # middle rows: back wall
    assert (floor > 0).mean() < 0.05
    assert (wall > 0).mean() > 0.5


# ---------------------------------------------------------------- resolve pass (screen.frag + RGBA8 read-back)
from conftest import GOLDEN, screen_cases  # noqa: E402


@pytest.mark.parametrize("case", screen_cases(), ids=lambda c: c[0])
def test_resolve_restatement_reproduces_reference_bytes(case):
    """oracle/pt_oracle.c rs_* == the reference's screen.frag drawn into RGBA8 on llvmpipe, byte for byte (power-of-two sizes)."""
    _, acc, gamma, want = case
    got = pt_oracle.resolve(acc, gamma)
    assert got.shape == want.shape and np.array_equal(got, want), f"{int((got != want).sum())} bytes differ"
    assert np.array_equal(pt_oracle.resolve(acc, gamma, flip_y=True), want[::-1])  # saveCurrentFrame's vertical flip (window.cpp:391-398)


def test_resolve_npot_fixture_documents_the_sampler_leak():
    """At a non-power-of-two size the reference's GL_LINEAR samplers let a neighbour texel leak in; on white noise --
    the worst case -- that moves the recorded number of bytes by a few LSB.  The exact-texel restatement must differ by exactly that."""
    z = np.load(GOLDEN / "screen_npot_50x38.npz")
    acc = np.concatenate([z["rgb"], z["count"][..., None]], -1).astype(np.float32)
    got = pt_oracle.resolve(acc, 2.2)
    d = np.abs(got.astype(int) - z["out"][0].astype(int))
    assert int((d != 0).sum()) == int(z["exact_texel_mismatching_bytes"]) and int((d != 0).sum()) <= 8 and d.max() <= 8
