"""glrt_bvh_build_reference -- the reference host's own tree, its builder restated rule for rule (src/core/bvh.cpp:72-160, bvh.h:11-82).

PARITY UNPINNED: the reference's host cannot be built here, so no tree of its making exists to compare with.  What is checked instead:
  * CPU: at EVERY fork of the built tree, a numpy statement of the same arithmetic (float32 where the reference computes in float, float64 where it widens) must choose
    the same axis and the same split -- the triangle SETS of the two children, for the nth_element case their order along the axis --, the node boxes and the pre-order
    numbering of bvh.cpp:77-78 / :155-157;
  * GPU: the HIP path on this tree equals the oracle on this tree bit for bit, on a scene with coincident faces of different materials (exact ties: the one place,
    with grazing-ray box misses, where the tree shows in the image), and the scene builders leave this tree's child order alone.
"""
import numpy as np
import pytest

from glrt_amd import host, scenes

F32 = np.float32


def _positions(sc):
    v = sc["vert"].reshape(-1, 5, 3)[:, 0]
    return v[sc["tri"][:, :3].astype(np.int64)]  # (nT, 3 vertices, 3)


def _area(lo, hi):
    s = np.abs(hi - lo).astype(F32)
    return F32(2.0) * F32(F32(F32(s[0] * s[1]) + F32(s[1] * s[2])) + F32(s[2] * s[0]))


def _union(lo, hi, members):
    """Box of a set of triangles, accumulated from +-1e8 like bvh.h:12-15 (min / max are exact: the order does not matter)."""
    if len(members) == 0:
        return np.full(3, 1.0e8, F32), np.full(3, -1.0e8, F32)
    return np.minimum(F32(1.0e8), lo[members].min(0)), np.maximum(F32(-1.0e8), hi[members].max(0))


def _expected_split(members, cen, lo, hi):
    """The reference's decision for the triangles `members` of one fork: (axis, kind, left set or None)."""
    k = len(members)
    c = cen[members]
    clo, chi = np.minimum(F32(1.0e8), c.min(0)), np.maximum(F32(-1.0e8), c.max(0))
    span = np.abs(chi - clo).astype(F32)
    m = max(span[0], max(span[1], span[2]))
    axis = 0 if m == span[0] else (1 if m == span[1] else 2)
    if k <= 8:
        return axis, "nth", None
    B = 16
    cmin, cmax = np.float64(clo[axis]), np.float64(chi[axis])
    idenom = 1.0 / (abs(cmax - cmin) + 1.0e-8)
    numer = (c[:, axis] - clo[axis]).astype(F32).astype(np.float64)  # the float difference, widened afterwards (:113)
    b = np.minimum((B * np.abs(numer) * idenom).astype(np.int64), B - 1)
    alo, ahi = _union(lo, hi, members)
    area_all = _area(alo, ahi)
    cost = np.zeros(B - 1)
    with np.errstate(all="ignore"):
        for i in range(B - 1):
            m0, m1 = members[b <= i], members[b > i]
            a0, a1 = _area(*_union(lo, hi, m0)), _area(*_union(lo, hi, m1))
            ratio = F32(F32(F32(len(m0)) * a0) + F32(F32(len(m1)) * a1)) / area_all
            cost[i] = 0.125 + np.float64(F32(ratio))
    split, best = 0, cost[0]
    for i in range(1, B - 1):
        if best > cost[i]:
            best, split = cost[i], i
    if not best < k:
        return axis, "middle", None
    diff = np.abs(c[:, axis].astype(np.float64) - cmin)  # the double difference of the partition's predicate (:43)
    b2 = np.minimum((B * diff * idenom).astype(np.int64), B - 1)
    left = set(members[b2 <= split].tolist())
    if len(left) in (0, k):
        return axis, "middle", None
    return axis, "partition", left


def _check_tree(sc, nodes):
    nodes = np.asarray(nodes, F32).reshape(-1, 9)
    p = _positions(sc)
    n_tri = p.shape[0]
    assert nodes.shape[0] == 2 * n_tri - 1
    lo, hi = p.min(1), p.max(1)
    cen = ((p[:, 0] + p[:, 1]).astype(F32) + p[:, 2]).astype(F32) / F32(3.0)
    kinds = {"nth": 0, "partition": 0, "middle": 0}

    def leaves(n):
        out, st = [], [n]
        while st:
            x = st.pop()
            if nodes[x, 8] >= 0:
                out.append(int(nodes[x, 8]))
            else:
                st.append(int(nodes[x, 7])); st.append(int(nodes[x, 6]))
        return np.asarray(out, np.int64)

    st, seen = [0], []
    while st:
        n = st.pop()
        mem = leaves(n)
        blo, bhi = _union(lo, hi, mem)
        assert np.array_equal(nodes[n, 0:3], blo) and np.array_equal(nodes[n, 3:6], bhi), f"node {n}: box"
        if nodes[n, 8] >= 0:
            assert nodes[n, 6] == -1 and nodes[n, 7] == -1
            seen.append(int(nodes[n, 8]))
            continue
        l, r = int(nodes[n, 6]), int(nodes[n, 7])
        L, R = leaves(l), leaves(r)
        assert l == n + 1 and r == n + 2 * len(L), f"node {n}: pre-order numbering"
        axis, kind, left = _expected_split(mem, cen, lo, hi)
        kinds[kind] += 1
        if kind == "partition":
            assert set(L.tolist()) == left, f"node {n}: the bucket split's left side"
        else:
            assert len(L) == len(mem) // 2, f"node {n}: cut at the middle"
            if kind == "nth":
                assert cen[L, axis].max() <= cen[R, axis].min(), f"node {n}: nth_element along axis {axis}"
        st.append(r); st.append(l)
    assert sorted(seen) == list(range(n_tri))
    return kinds


def _build(sc):
    nodes, depth = host.build_bvh(sc["vert"], sc["tri"], "reference")
    return nodes, depth


@pytest.mark.parametrize("make", [
    lambda: scenes.config_c1(16, 16, bvh="sah", subdiv=1)[0],                 # spheres + quads: many equal centroids' spans, flat boxes
    lambda: scenes.config_c2(16, 16, bvh="sah", subdiv=1)[0],                 # the Cornell box of the headline, small
    lambda: scenes.config_c3(16, 16, n=700, bvh="sah")[0],                    # random triangles
    lambda: scenes.config_c3(16, 16, n=9, bvh="sah")[0],                      # one bucket level above the nth_element levels
    lambda: scenes.config_c3(16, 16, n=2, bvh="sah")[0],
    lambda: scenes.config_c3(16, 16, n=1, bvh="sah")[0],
], ids=["c1", "c2", "random700", "random9", "two", "one"])
def test_every_fork_follows_the_reference_builders_rule(make):
    sc = make()
    nodes, depth = _build(sc)
    kinds = _check_tree(sc, nodes)
    n = sc["tri"].shape[0]
    assert sum(kinds.values()) == n - 1
    if n > 8:
        assert kinds["partition"] > 0
    assert 0 <= depth < 63


def test_equal_centroids_are_cut_at_the_unsorted_middle():
    """Triangles whose centroids all coincide land in one bucket: no split costs less than the count (:147), and the reference cuts the range at its middle in the order
    it has -- without sorting it.  With the input order 0..n-1 the tree's leaves are then 0..n-1 from left to right down to the nth_element levels."""
    b = scenes.SceneBuilder()
    m = b.add_material(scenes.diffuse((0.5, 0.5, 0.5)))
    rng = np.random.default_rng(5)
    pos = []
    for i in range(40):  # 40 triangles around one common centroid (vertices v, w, -(v + w))
        v, w = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
        pos.append([v, w, -(v + w)])
    pos = np.asarray(pos, F32)
    # make the centroids EXACTLY equal: overwrite the third vertex so that ((v0 + v1) + v2) / 3 rounds to 0 in float32
    pos[:, 2] = -(pos[:, 0] + pos[:, 1]).astype(F32)
    b.add_mesh(pos, np.tile(np.asarray([0, 0, 1], F32), (40, 3, 1)), m)
    sc = b.build("sah")
    nodes, _ = _build(sc)
    kinds = _check_tree(sc, nodes)
    assert kinds["middle"] >= 1 and kinds["partition"] == 0
    nodes = np.asarray(nodes, F32).reshape(-1, 9)
    # the root and its two children cut [0, 40) at 20 and [0, 20) / [20, 40) at 10 / 30 without reordering
    def tri_set(n):
        out, st = set(), [n]
        while st:
            x = st.pop()
            if nodes[x, 8] >= 0: out.add(int(nodes[x, 8]))
            else: st.extend([int(nodes[x, 6]), int(nodes[x, 7])])
        return out
    l, r = int(nodes[0, 6]), int(nodes[0, 7])
    assert tri_set(l) == set(range(0, 20)) and tri_set(r) == set(range(20, 40))
    assert tri_set(int(nodes[l, 6])) == set(range(0, 10)) and tri_set(int(nodes[r, 7])) == set(range(30, 40))


def test_the_scene_builders_leave_the_reference_tree_in_its_own_order():
    """Every other builder's tree is handed out with the light side first (glrt_bvh_lights_first); this one's point is the reference's own visiting order."""
    sc, _ = scenes.config_c2(16, 16, bvh="reference", subdiv=1)
    assert sc["bvh_lights_first"] == 0 and np.array_equal(sc["bvh"], sc["bvh_builder"])
    sc2 = scenes.rebuild_bvh(scenes.config_c2(16, 16, bvh="sah", subdiv=1)[0], "reference")
    assert sc2["bvh_lights_first"] == 0 and np.array_equal(sc2["bvh"], sc["bvh"])


def test_bad_arguments_are_refused():
    sc, _ = scenes.config_c3(16, 16, n=5, bvh="sah")
    tri = sc["tri"].copy()
    tri[2, 1] = 1e9  # a vertex index out of range
    with pytest.raises(RuntimeError):
        host.build_bvh(sc["vert"], tri, "reference")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["reference", "sah"])
def test_gpu_equals_the_oracle_on_coincident_faces_under_the_reference_tree(gpu_device, kind):
    """Exact ties are where the tree shows: every quad of the box duplicated in place with another material id.  Device and oracle walk the same tree in the same order,
    so they agree bit for bit whichever tree it is; the two trees' images differ from each other at tied pixels (that difference is the reason the builder exists)."""
    from oracle import pt_oracle
    from conftest import assert_bit_equal
    sc0, params = scenes.config_c2(96, 64, max_depth=4, bvh="sah", subdiv=1)
    p = _positions(sc0)
    vert = sc0["vert"].reshape(-1, 5, 3)
    tri = sc0["tri"]
    n_mat = sc0["mat"].reshape(-1, 6, 3).shape[0]
    # duplicate every triangle whose box is flat (the walls' quads) with the material id rotated by one
    flat = (np.abs(p.max(1) - p.min(1)).min(1) == 0)
    dup = np.nonzero(flat)[0]
    assert len(dup) >= 10
    extra_vert = vert[tri[dup, :3].astype(np.int64).reshape(-1)]
    base = vert.shape[0]
    extra_tri = np.concatenate([np.arange(base, base + 3 * len(dup), dtype=F32).reshape(-1, 3), ((tri[dup, 3] + 1) % n_mat)[:, None]], 1).astype(F32)
    vert2 = np.concatenate([vert, extra_vert], 0).reshape(-1, 3)
    tri2 = np.concatenate([tri, extra_tri], 0).astype(F32)
    sc = dict(sc0, vert=vert2, tri=tri2)
    sc = scenes.rebuild_bvh(sc, kind)
    emis = sc["mat"].reshape(-1, 6, 3)[tri2[:, 3].astype(np.int64), 0]
    lights = np.nonzero(np.linalg.norm(emis.astype(np.float64), axis=1) != 0.0)[0]
    sc["light"] = np.ascontiguousarray(tri2[lights].reshape(-1, 4))
    ref, ref_rays = pt_oracle.render(sc, params)
    d = gpu_device
    d.upload_scene(sc); d.set_partition(0, 1, 16); d.resize(params["width"], params["height"]); d.reset_stats(); d.count_rays(True)
    d.render(params); d.sync()
    acc, st = d.read_accum(), d.stats()
    assert st.rays == ref_rays
    assert_bit_equal(acc, ref, f"{kind} tree, coincident faces")
