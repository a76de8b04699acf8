"""Host logic: BVH builders, camera matrices, scene validation/repacking, row partition."""
import ctypes as C

import numpy as np
import pytest

from glrt_amd import device, dist, host, scenes


def _nodes(scene):
    return np.asarray(scene["bvh"], np.float32).reshape(-1, 9)


def _tri_boxes(scene):
    v = scene["vert"].reshape(-1, 5, 3)[:, 0]
    idx = scene["tri"][:, :3].astype(np.int64)
    p = v[idx]
    return p.min(1), p.max(1)


@pytest.mark.parametrize("kind", ["sah", "chain", "lbvh", "sahl", "reference"])
def test_bvh_is_a_valid_tree_over_all_triangles(kind):
    sc, _ = scenes.config_c3(32, 32, n=700, bvh=kind)
    nodes = _nodes(sc)
    n_tri = sc["tri"].shape[0]
    assert nodes.shape[0] == 2 * n_tri - 1
    lo, hi = _tri_boxes(sc)
    seen_tri = np.zeros(n_tri, bool)
    seen_node = np.zeros(nodes.shape[0], bool)
    stack = [(0, None)]
    while stack:
        n, parent_box = stack.pop()
        assert not seen_node[n]
        seen_node[n] = True
        bmin, bmax, ch = nodes[n, 0:3], nodes[n, 3:6], nodes[n, 6:9]
        if parent_box is not None and kind != "chain":
            assert np.all(bmin >= parent_box[0]) and np.all(bmax <= parent_box[1])
        if ch[2] < 0:
            for c in ch[:2]:
                assert c >= 0
                stack.append((int(c), (bmin, bmax)))
        else:
            t = int(ch[2])
            assert ch[0] == -1 and ch[1] == -1 and not seen_tri[t]
            seen_tri[t] = True
            assert np.all(bmin <= lo[t]) and np.all(bmax >= hi[t])
    assert seen_tri.all() and seen_node.all()


def test_chain_bvh_visits_triangles_in_order_with_reference_traversal():
    sc, _ = scenes.config_c3(16, 16, n=50, bvh="chain")
    nodes = _nodes(sc)
    order, stack, depth = [], [0], 0
    while stack:  # push children.x then .y, pop .y first (raytrace.frag:299-307)
        n = stack.pop()
        ch = nodes[n, 6:9]
        if ch[2] < 0:
            stack.extend(int(c) for c in ch[:2] if c >= 0)
            depth = max(depth, len(stack))
        else:
            order.append(int(ch[2]))
    assert order == list(range(50)) and depth == 2


def test_lbvh_layout_and_image_equals_sah_image():
    """LBVH: internal node i at index i, leaves after them in Morton order (unique keys => strictly increasing);
    the oracle renders the same image with either tree (random triangles: no exact ties)."""
    from oracle import pt_oracle
    sc_l, pr = scenes.config_c3(48, 32, n=900, bvh="lbvh", max_depth=3)
    sc_s, _ = scenes.config_c3(48, 32, n=900, bvh="sah", max_depth=3)
    nodes = _nodes(sc_l)
    n = sc_l["tri"].shape[0]
    assert np.all(nodes[:n - 1, 8] == -1.0) and np.all(nodes[n - 1:, 8] >= 0.0)
    assert sorted(nodes[n - 1:, 8].astype(int).tolist()) == list(range(n))
    assert 10 <= sc_l["bvh_depth"] < 63
    img_l, rays_l = pt_oracle.render(sc_l, pr)
    img_s, rays_s = pt_oracle.render(sc_s, pr)
    assert rays_l == rays_s
    assert np.array_equal(img_l.view(np.uint32), img_s.view(np.uint32))
    # degenerate inputs: one triangle, two triangles, many identical triangles (equal Morton codes)
    for k in (1, 2, 3):
        sc1, _ = scenes.config_c3(8, 8, n=k, bvh="lbvh")
        assert _nodes(sc1).shape[0] == 2 * k - 1
    v = np.tile(sc_l["vert"].reshape(-1, 5, 3)[:3], (40, 1, 1)).reshape(-1, 3)
    t = np.array([[3 * i, 3 * i + 1, 3 * i + 2, 0] for i in range(40)], np.float32)
    nodes40, depth40 = host.build_bvh(v, t, "lbvh")
    assert depth40 <= 8  # index tie-break splits equal codes evenly


def _tree_stats(nodes):
    """(leaf count, forks' summed half-area / root's, summed half-area of the leaves' parents / root's, depth); asserts a proper tree with tight boxes."""
    n_nodes = nodes.shape[0]
    seen = np.zeros(n_nodes, bool)
    area = lambda b: (lambda d: d[0] * d[1] + d[1] * d[2] + d[2] * d[0])(np.maximum(b[3:6] - b[0:3], 0).astype(np.float64))
    forks = tests = 0.0
    leaves = depth = 0
    st = [(0, 0)]
    while st:
        i, d = st.pop()
        assert not seen[i]
        seen[i] = True
        if nodes[i, 8] >= 0:
            leaves += 1; depth = max(depth, d)
            continue
        l, r = int(nodes[i, 6]), int(nodes[i, 7])
        assert np.array_equal(nodes[i, 0:3], np.minimum(nodes[l, 0:3], nodes[r, 0:3]))  # the union of the children, exactly
        assert np.array_equal(nodes[i, 3:6], np.maximum(nodes[l, 3:6], nodes[r, 3:6]))
        forks += area(nodes[i])
        tests += sum(area(nodes[i]) for c in (l, r) if nodes[c, 8] >= 0)
        st += [(l, d + 1), (r, d + 1)]
    assert seen.all()
    return leaves, forks / area(nodes[0]), tests / area(nodes[0]), depth


def test_lbvh_quality_passes_keep_a_valid_tree_and_reach_the_sah_trees_cost():
    """Rotation sweeps + SAH rebuild of the subtrees of <= GLRT_LBVH_REBUILD_LEAVES leaves (host/bvh.cpp): every triangle once, every
    fork's box the exact union of its children's, and on random triangles the surface-area cost lands within 3 % of the
    binned-SAH tree's (Morton tree + rotations alone: 6 % more expected triangle tests)."""
    sc_l, _ = scenes.config_c3(16, 16, n=20_000, bvh="lbvh")
    sc_s, _ = scenes.config_c3(16, 16, n=20_000, bvh="sah")
    n_l, forks_l, tests_l, depth_l = _tree_stats(_nodes(sc_l))
    n_s, forks_s, tests_s, _ = _tree_stats(_nodes(sc_s))
    assert n_l == n_s == 20_000 and depth_l == sc_l["bvh_depth"] < 40
    assert forks_l < 1.03 * forks_s and tests_l < 1.03 * tests_s
    # equal boxes: every split costs the same; the tie goes to the balanced one, not to a chain
    v = np.tile(sc_l["vert"].reshape(-1, 5, 3)[:3], (64, 1, 1)).reshape(-1, 3)
    t = np.array([[3 * i, 3 * i + 1, 3 * i + 2, 0] for i in range(64)], np.float32)
    nodes, depth = host.build_bvh(v, t, "lbvh")
    assert _tree_stats(nodes.reshape(-1, 9))[0] == 64 and depth == 6
    # a box coordinate of -0.0 next to +0.0: unions are formed without negative zeros
    sc, _ = scenes.config_c3(8, 8, n=50, bvh="sah")
    vert = sc["vert"].reshape(-1, 5, 3).copy()
    vert[::7, 0, 1] = -0.0
    vert[3::7, 0, 1] = 0.0
    nodes, _ = host.build_bvh(vert.reshape(-1, 3), sc["tri"], "lbvh")
    nodes = nodes.reshape(-1, 9)
    forks = nodes[nodes[:, 8] < 0]
    assert not np.any((forks[:, :6] == 0) & np.signbit(forks[:, :6]))


def test_sah_depth_bounded():
    sc, _ = scenes.config_c2(32, 32, subdiv=2)
    assert sc["bvh_depth"] < 40


def test_camera_matrices_match_numpy():
    eye, tgt, up = np.array([0, 3, 9.0]), np.array([0, 1, 0.0]), np.array([0, 1, 0.0])
    v = host.look_at(eye, tgt, up).reshape(4, 4).T  # to row-major
    f = (tgt - eye) / np.linalg.norm(tgt - eye)
    s = np.cross(f, up); s /= np.linalg.norm(s)
    u = np.cross(s, f)
    ref = np.eye(4)
    ref[0, :3], ref[1, :3], ref[2, :3] = s, u, -f
    ref[:3, 3] = [-s @ eye, -u @ eye, f @ eye]
    assert np.allclose(v, ref, atol=1e-6)
    p = host.perspective(40.0, 16 / 9, 0.1, 100.0).reshape(4, 4).T
    t = np.tan(np.radians(40.0) / 2)
    assert np.isclose(p[0, 0], 1 / (16 / 9 * t), rtol=1e-6) and np.isclose(p[1, 1], 1 / t, rtol=1e-6)
    assert np.isclose(p[3, 2], -1.0) and np.isclose(p[2, 3], -2 * 100 * 0.1 / (100 - 0.1), rtol=1e-6)
    inv = host.mat4_inverse(host.look_at(eye, tgt, up)).reshape(4, 4).T
    assert np.allclose(inv @ v, np.eye(4), atol=1e-5)
    assert np.allclose(host.mat4_mul(host.look_at(eye, tgt, up), host.mat4_inverse(host.look_at(eye, tgt, up))).reshape(4, 4),
                       np.eye(4), atol=1e-5)


def test_frame_seeds_in_unit_interval():
    s = [host.frame_seed(f) for f in range(100)]
    assert all(0 <= a < 1 and 0 <= b < 1 for a, b in s) and len(set(s)) == 100
    assert s[0] == (float(np.float32(0.137)), float(np.float32(0.731)))


# ---- scene validation / repacking (glrtx_check_scene: host-only entry of the device library)
def _check(sc):
    L = device.lib()
    fp = C.POINTER(C.c_float)
    L.glrtx_check_scene.argtypes = [fp, C.c_size_t] * 5 + [C.POINTER(C.c_int)] * 2
    arrs = [np.ascontiguousarray(sc[k], np.float32) for k in ("vert", "tri", "mat", "light", "bvh")]
    per = (15, 4, 18, 4, 9)
    args = []
    for a, n in zip(arrs, per):
        args += [a.ctypes.data_as(fp), a.size // n]
    nf, se = C.c_int(), C.c_int()
    rc = L.glrtx_check_scene(*args, C.byref(nf), C.byref(se))
    return rc, nf.value, se.value, L.glrtx_last_error(None).decode()


def _leaf_pairs(sc):
    """Forks of the wire tree whose two children are both leaves: packed as two chained triangle records, without a fork record."""
    n = sc["bvh"].reshape(-1, 9)
    fork = n[:, 8] < 0
    f = np.flatnonzero(fork)
    l, r = n[f, 6].astype(int), n[f, 7].astype(int)
    return int(((l >= 0) & (r >= 0) & ~fork[np.maximum(l, 0)] & ~fork[np.maximum(r, 0)]).sum())


def test_check_scene_accepts_generated_scenes():
    for name in ("c1", "c2"):
        sc, _ = scenes.CONFIGS[name](32, 32)
        rc, n_fork, need, _ = _check(sc)
        assert rc == 0 and n_fork == sc["tri"].shape[0] - 1 - _leaf_pairs(sc) and 0 < need <= sc["bvh_depth"]
        assert _leaf_pairs(sc) > 0.3 * sc["tri"].shape[0] / 2  # the SAH builder ends most branches in a pair of leaves
    sc, _ = scenes.config_c3(32, 32, n=300, bvh="chain")
    assert _check(sc)[:3] == (0, 298, 1)  # 299 forks, the last one holds two leaves


def test_leaf_pairs_are_chained_triangle_records():
    """A fork with two leaf children leaves no fork record: its parent refers to the triangle record of children.y (popped first,
    raytrace.frag:299-307), which names the record of children.x -- its predecessor in memory.  Checked on the packed refs."""
    sc, _ = scenes.config_c1(16, 16, subdiv=1)
    rc, forks, root, need = _pack(sc)
    assert rc == 0
    refs = forks.view(np.int32)[:, [3, 7]]
    n_tri = sc["tri"].reshape(-1, 4).shape[0]
    leaf_refs = refs[refs < -1]
    assert len(forks) == n_tri - 1 - _leaf_pairs(sc)
    # every triangle id 1..n_tri is reachable: directly from a fork, or as the second record of a pair (first - 1)
    ids = set((~leaf_refs).tolist())
    assert len(ids) == len(leaf_refs)
    assert len(ids) + _leaf_pairs(sc) == n_tri and max(ids) <= n_tri and min(ids) >= 1
    # two-triangle scene: the root itself is a pair -> no fork at all, the root ref is a triangle record and the root box is still tested
    sc2, _ = scenes.config_c3(16, 16, n=2, bvh="chain")
    rc, forks2, root2, need2 = _pack(sc2)
    assert rc == 0 and len(forks2) == 0 and root2 == ~1 and need2 == 0


def test_check_scene_rejects_malformed_scenes():
    sc, _ = scenes.config_c1(32, 32, subdiv=1)
    bad = dict(sc, tri=sc["tri"].copy()); bad["tri"][5, 1] = 1e9
    assert _check(bad)[0] == device.GLRTX_ESCENE
    bad = dict(sc, tri=sc["tri"].copy()); bad["tri"][7, 3] = 99
    assert _check(bad)[0] == device.GLRTX_ESCENE
    bad = dict(sc, bvh=sc["bvh"].copy()); bad["bvh"].reshape(-1, 9)[0, 6] = 0  # root's left child = root: a cycle
    rc, _, _, msg = _check(bad)
    assert rc == device.GLRTX_ESCENE and "more than once" in msg
    bad = dict(sc, bvh=sc["bvh"].copy()); bad["bvh"].reshape(-1, 9)[0, 7] = 1e7
    assert _check(bad)[0] == device.GLRTX_ESCENE


def test_check_scene_rejects_too_deep_tree():
    # a right-leaning comb: every fork = (leaf, next fork) keeps one entry per level on the stack
    n = 80
    sc, _ = scenes.config_c3(16, 16, n=n, bvh="chain")
    nodes = sc["bvh"].reshape(-1, 9).copy()
    forks = nodes[nodes[:, 8] < 0]
    for f in range(nodes.shape[0]):
        if nodes[f, 8] < 0:
            nodes[f, 6], nodes[f, 7] = nodes[f, 7], nodes[f, 6]  # leaf becomes children.x (stacked), chain continues in .y
    rc, _, _, msg = _check(dict(sc, bvh=nodes))
    assert rc == device.GLRTX_EDEPTH and "stack" in msg
    assert len(forks) == n - 1


def _pack(sc):
    """glrtx_debug_pack_forks: (rc, fork records (n, 16) float32, root ref, stack entries)."""
    L = device.lib()
    fp = C.POINTER(C.c_float)
    L.glrtx_debug_pack_forks.argtypes = [fp, C.c_size_t] * 5 + [fp, C.c_size_t] + [C.POINTER(C.c_int)] * 3
    arrs = [np.ascontiguousarray(sc[k], np.float32) for k in ("vert", "tri", "mat", "light", "bvh")]
    args = []
    for a, n in zip(arrs, (15, 4, 18, 4, 9)):
        args += [a.ctypes.data_as(fp), a.size // n]
    nf, root, se = C.c_int(), C.c_int(), C.c_int()
    rc = L.glrtx_debug_pack_forks(*args, None, 0, C.byref(nf), C.byref(root), C.byref(se))
    if rc != 0:
        return rc, None, 0, 0
    forks = np.zeros((max(nf.value, 1), 16), np.float32)
    rc = L.glrtx_debug_pack_forks(*args, forks.ctypes.data_as(fp), forks.shape[0], C.byref(nf), C.byref(root), C.byref(se))
    return rc, forks[:nf.value], root.value, se.value


def _deepest_stack(forks, root):
    """Replay of trav_step's push / pop rules (pt_kernel.hip.h) with every box test passing -- the worst case, a failed test only
    removes entries: at a fork both children "pass" (a leaf or an absent child always does), the left ref is pushed, the walk goes on
    with the right one; a leaf (ref < 0, the never-hit record included) pops.  Returns (deepest stack, records visited)."""
    refs = forks.view(np.int32)[:, [3, 7]]
    cur, stack, deepest, visited = root, [], 0, 0
    while True:
        visited += 1
        if cur >= 0:
            stack.append(int(refs[cur, 0]))
            deepest = max(deepest, len(stack))
            cur = int(refs[cur, 1])
        elif stack:
            cur = stack.pop()
        else:
            return deepest, visited


def _with_one_child_forks(scene, n_extra, seed, chain=1):
    """Put `chain` one-child forks (children.x or children.y left out) in front of n_extra children."""
    nodes = scene["bvh"].reshape(-1, 9).copy()
    rng = np.random.default_rng(seed)
    forks = np.flatnonzero(nodes[:, 8] < 0)
    extra = []
    for k, f in enumerate(rng.choice(forks, min(n_extra, len(forks)), replace=False)):
        side = 6 + (k & 1)
        child = int(nodes[f, side])
        box = nodes[child, 0:6].copy()            # every inserted fork has the box of the child the chain leads to
        for j in range(chain):
            unary = np.zeros(9, np.float32)
            unary[0:6] = box
            unary[6:9] = (-1.0, -1.0, -1.0)
            unary[6 + (((k + j) >> 1) & 1)] = float(child)   # hanging off children.x for some, children.y for others
            child = nodes.shape[0] + len(extra)
            extra.append(unary)
        nodes[f, side] = float(child)
    return dict(scene, bvh=np.concatenate([nodes, np.array(extra, np.float32).reshape(-1, 9)], 0).reshape(-1, 3))


@pytest.mark.parametrize("case", ["sah", "lbvh", "sahl", "comb", "one_child", "one_child_chains", "single_triangle", "empty"])
def test_stack_entries_cover_the_deepest_stack_the_traversal_step_can_reach(case):
    """an absent child is the never-hit record and IS pushed; the stack budget must count it."""
    if case == "empty":
        sc = dict(vert=np.zeros((0, 3), np.float32), tri=np.zeros((0, 4), np.float32), mat=np.zeros((6, 3), np.float32),
                  light=np.zeros((0, 4), np.float32), bvh=np.zeros((0, 3), np.float32))
    elif case == "single_triangle":
        sc, _ = scenes.config_c3(16, 16, n=1)
    elif case == "comb":
        sc, _ = scenes.config_c3(16, 16, n=60, bvh="chain")
        nodes = sc["bvh"].reshape(-1, 9).copy()
        fk = nodes[:, 8] < 0
        nodes[fk, 6], nodes[fk, 7] = nodes[fk, 7].copy(), nodes[fk, 6].copy()
        sc = dict(sc, bvh=nodes.reshape(-1, 3))
    elif case == "lbvh":
        sc, _ = scenes.config_c3(16, 16, n=3000, bvh="lbvh")
    elif case == "sahl":
        sc, _ = scenes.config_c3(16, 16, n=3000, bvh="sahl")
    else:
        sc, _ = scenes.config_c1(16, 16, subdiv=1)
        if case == "one_child":
            sc = _with_one_child_forks(sc, 60, 5)
        elif case == "one_child_chains":
            sc = _with_one_child_forks(sc, 25, 7, chain=5)
    rc, forks, root, need = _pack(sc)
    assert rc == 0
    n_tri = sc["tri"].reshape(-1, 4).shape[0]
    if root < 0:
        assert need == 0 and n_tri == 1
        return
    deepest, visited = _deepest_stack(forks, root)
    assert deepest == need, (case, deepest, need)
    refs = forks.view(np.int32)[:, [3, 7]]
    assert (refs >= ~n_tri).all() and (refs < len(forks)).all()
    assert visited >= 2 * len(forks)  # every fork and both of its refs were met once: the packed refs form a tree


def test_a_long_chain_of_one_child_forks_needs_one_stack_entry_like_the_reference():
    """raytrace.frag:299-307 pushes only the children that exist, so a chain of one-child forks keeps its stack at one entry; the
    packed tree stores an only child in the left slot, and neither overflows the budget nor trips GLRTX_EDEPTH."""
    sc, _ = scenes.config_c3(16, 16, n=2, bvh="chain")
    nodes = sc["bvh"].reshape(-1, 9)                # node 0: a fork over two leaves (nodes 1 and 2)
    assert nodes.shape[0] == 3 and nodes[0, 8] < 0
    n_chain = 100
    out = [None] * (3 + n_chain)
    out[1], out[2] = nodes[1].copy(), nodes[2].copy()
    out[3 + n_chain - 1] = nodes[0].copy()          # the two-leaf fork, at the end of the chain
    for j in range(n_chain):                        # node 0, 3, 4, ...: one-child forks, each leading to the next
        u = nodes[0].copy()
        u[6:9] = (-1.0, -1.0, -1.0)
        u[6 + (j & 1)] = float(3 + j)
        out[0 if j == 0 else 3 + j - 1] = u
    rc, forks, root, need = _pack(dict(sc, bvh=np.array(out, np.float32).reshape(-1, 3)))
    assert rc == 0, device.lib().glrtx_last_error(None)
    assert len(forks) == n_chain  # the two-leaf fork at the end of the chain is packed as a pair of chained triangle records
    deepest, _ = _deepest_stack(forks, root)
    assert need == deepest == 1


def test_empty_scene_is_valid():
    sc = dict(vert=np.zeros((0, 3), np.float32), tri=np.zeros((0, 4), np.float32), mat=np.zeros((6, 3), np.float32),
              light=np.zeros((0, 4), np.float32), bvh=np.zeros((0, 3), np.float32))
    assert _check(sc)[0] == 0


# ---- partition
@pytest.mark.parametrize("world,stripe,height", [(1, 16, 1080), (2, 16, 1080), (8, 16, 1080), (8, 16, 2160), (3, 32, 100), (8, 16, 40)])
def test_row_partition_is_a_bijection(world, stripe, height):
    allrows = np.concatenate([dist.owned_rows(r, world, stripe, height) for r in range(world)])
    assert sorted(allrows.tolist()) == list(range(height))
    for r in range(world):
        ys = dist.owned_rows(r, world, stripe, height)
        for i, y in enumerate(ys):  # same formula as the kernel / glrtx_local_row_to_y
            assert y == ((i // stripe) * world + r) * stripe + i % stripe


@pytest.mark.parametrize("kind", ["sah", "chain", "lbvh", "sahl", "reference"])
def test_builders_take_non_finite_vertices(kind):
    """Vertices at +-inf, at 3e38 (extents that overflow) and NaN: every builder returns a tree over all triangles that the device layer accepts.  A box's
    centre that is not finite is ordered and binned as 0 (host/bvh.cpp: centre, bin_of); until round 4 the SAH builder indexed its bins with (int)NaN."""
    sc0, _ = scenes.config_c1(16, 16, bvh="sah", subdiv=1)
    for val in (np.nan, np.inf, -np.inf, 3e38, None):
        vert = sc0["vert"].reshape(-1, 5, 3).copy()
        if val is None:
            vert[7, 0, 1], vert[8, 0, 1], vert[100, 0, 0], vert[101, 0, 0], vert[333, 0, 2] = np.inf, -np.inf, 3e38, -3e38, np.nan
        else:
            vert[7, 0, 1] = vert[100, 0, 0] = vert[333, 0, 2] = val
        sc = scenes.rebuild_bvh(dict(sc0, vert=vert.reshape(-1, 3)), kind)
        nodes = _nodes(sc)
        n_tri = sc["tri"].shape[0]
        assert nodes.shape[0] == 2 * n_tri - 1
        leaves = nodes[nodes[:, 8] >= 0]
        assert sorted(leaves[:, 8].astype(int).tolist()) == list(range(n_tri)), "every triangle once"
        rc, _, _, msg = _check(sc)
        assert rc == device.GLRTX_OK, msg


def _tree_ok(nodes, n_tri):
    """Every node reached exactly once from the root, n_tri leaves, every fork's box the union of its children's (as float sets: min / max)."""
    N = np.asarray(nodes, np.float32).reshape(-1, 9)
    seen = np.zeros(N.shape[0], np.int64)
    tris = []
    st = [0]
    while st:
        i = st.pop()
        seen[i] += 1
        if N[i, 8] < 0:
            l, r = int(N[i, 6]), int(N[i, 7])
            lo = np.minimum(N[l, 0:3], N[r, 0:3]); hi = np.maximum(N[l, 3:6], N[r, 3:6])
            assert np.array_equal(lo + 0.0, N[i, 0:3] + 0.0) and np.array_equal(hi + 0.0, N[i, 3:6] + 0.0), i
            st += [l, r]
        else:
            tris.append(int(N[i, 8]))
    assert (seen == 1).all() and sorted(tris) == list(range(n_tri))


@pytest.mark.parametrize("n", [1, 2, 3, 64, 65, 66, 129, 777, 5000])
def test_sah_by_levels_builds_a_valid_tree_of_any_size(n):
    """glrt_bvh_build_sah_levels (round 5): binned SAH level by level down to segments of <= 64 triangles, the exact sweep SAH below; sizes around the 64-leaf
    boundary, where the root segment itself is closed or splits into a leaf and a closed subtree."""
    sc, _ = scenes.config_c3(8, 8, n=n, bvh="sah")
    nodes, depth = host.build_bvh(sc["vert"], sc["tri"], "sahl")
    _tree_ok(nodes, n)
    assert depth < 40
    again, _ = host.build_bvh(sc["vert"], sc["tri"], "sahl")
    assert np.array_equal(np.asarray(nodes).view(np.uint32), np.asarray(again).view(np.uint32))


def test_sah_by_levels_splits_by_position_where_no_bin_boundary_separates():
    """500 identical triangles: every centre in one bin on every axis, so every open segment is split by position (16 bins over its positions, the most even boundary):
    the tree stays shallow, and the image under it is the SAH tree's."""
    from oracle import pt_oracle
    sc, pr = scenes.config_c3(24, 16, n=3, bvh="sah", max_depth=2)
    v = np.tile(sc["vert"].reshape(-1, 5, 3)[:3], (500, 1, 1)).reshape(-1, 3)
    t = np.array([[3 * i, 3 * i + 1, 3 * i + 2, 0] for i in range(500)], np.float32)
    nodes, depth = host.build_bvh(v, t, "sahl")
    _tree_ok(nodes, 500)
    assert depth <= 14
    sah, _ = host.build_bvh(v, t, "sah")
    a, _ = pt_oracle.render(dict(sc, vert=v, tri=t, bvh=nodes, light=np.zeros((0, 4), np.float32)), pr)
    b, _ = pt_oracle.render(dict(sc, vert=v, tri=t, bvh=sah, light=np.zeros((0, 4), np.float32)), pr)
    assert np.array_equal(a[..., 3], b[..., 3]) and np.isfinite(a).all()


def test_sah_by_levels_reaches_the_binned_sah_trees_cost():
    """Summed surface area of the forks over the root's (expected fork visits of a random ray): within 1 % of the CPU binned-SAH tree's on config 5's kind of scene
    (the Morton tree with rotations and 64-leaf rebuilds is 0.7 % behind by this metric and 3.5 % behind in render time: profiles/r05_tree_study.txt)."""
    def cost(nodes):
        N = np.asarray(nodes, np.float32).reshape(-1, 9)
        e = N[:, 3:6] - N[:, 0:3]
        area = e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0]
        return float(area[N[:, 8] < 0].sum() / area[0])
    sc, _ = scenes.config_c5(16, 16, n=20_000)
    sahl, _ = host.build_bvh(sc["vert"], sc["tri"], "sahl")
    assert cost(sahl) <= 1.01 * cost(sc["bvh"])


def test_lights_first_exchanges_child_references_only_and_is_what_the_scene_builder_hands_out():
    """glrt_bvh_lights_first (round 5): at every fork where exactly one child's subtree holds emitting triangles, that child sits in the slot the reference's traversal
    visits first (children.y); boxes, leaves and the set of parent-child links are untouched; applying it again changes nothing; the oracle's image under the ordered tree
    is the image under the builder's tree (no exact ties in this scene); a chain (brute force) tree is left alone; GLRT_BVH_LIGHTS_FIRST=0 switches the pass off."""
    from oracle import pt_oracle
    for cfg, kw in (("c2", dict(width=48, height=32, subdiv=1)), ("c1", dict(width=48, height=32, subdiv=1)), ("c5", dict(width=32, height=24, n=3000))):
        sc, pr = scenes.CONFIGS[cfg](**kw)
        raw = np.asarray(sc["bvh_builder"], np.float32).reshape(-1, 9)
        got = np.asarray(sc["bvh"], np.float32).reshape(-1, 9)
        assert np.array_equal(raw[:, :6].view(np.uint32), got[:, :6].view(np.uint32)) and np.array_equal(raw[:, 8], got[:, 8])
        assert np.array_equal(np.sort(raw[:, 6:8], 1), np.sort(got[:, 6:8], 1))
        emits = np.linalg.norm(np.asarray(sc["mat"], np.float32).reshape(-1, 6, 3)[:, 1], axis=1) != 0
        tri_light = emits[np.asarray(sc["tri"], np.float32).reshape(-1, 4)[:, 3].astype(int)]
        has = np.zeros(got.shape[0], bool)
        order, st = [], [0]
        while st:
            i = st.pop(); order.append(i)
            if got[i, 8] < 0: st += [int(got[i, 6]), int(got[i, 7])]
        for i in reversed(order):
            has[i] = tri_light[int(got[i, 8])] if got[i, 8] >= 0 else has[int(got[i, 6])] or has[int(got[i, 7])]
        forks = [i for i in order if got[i, 8] < 0]
        assert not any(has[int(got[i, 6])] and not has[int(got[i, 7])] for i in forks), "a fork still holds its only light side in the x slot"
        assert sc["bvh_lights_first"] == int((raw[:, 6] != got[:, 6]).sum())
        again, n2 = host.lights_first(sc["bvh"], sc["tri"], sc["mat"])
        assert n2 == 0 and np.array_equal(np.asarray(again).view(np.uint32), np.asarray(sc["bvh"], np.float32).reshape(-1, 3).view(np.uint32))
        a, ra = pt_oracle.render(sc, pr)
        b, rb = pt_oracle.render(dict(sc, bvh=sc["bvh_builder"]), pr)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and ra == rb
    chain, _ = scenes.config_c3(8, 8, n=50, bvh="chain")
    assert chain["bvh_lights_first"] == 0 and np.array_equal(np.asarray(chain["bvh"]), np.asarray(chain["bvh_builder"]))


def _tree_facts(nodes):
    """(fork area sum / root area, depth, sorted leaf triangles, every child's box inside its parent's, node count reached from the root)"""
    N = np.asarray(nodes, np.float32).reshape(-1, 9)
    def area(i):
        d = N[i, 3:6].astype(np.float64) - N[i, :3]
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0]
    total, depth, leaves, nested, seen = 0.0, 0, [], True, 0
    st = [(0, 0)]
    while st:
        i, d = st.pop(); seen += 1; depth = max(depth, d)
        if N[i, 8] >= 0:
            leaves.append(int(N[i, 8])); continue
        total += area(i)
        for c in (int(N[i, 6]), int(N[i, 7])):
            nested &= bool((N[c, :3] >= N[i, :3]).all() and (N[c, 3:6] <= N[i, 3:6]).all())
            st.append((c, d + 1))
        lo, hi = np.minimum(N[int(N[i, 6]), :3], N[int(N[i, 7]), :3]), np.maximum(N[int(N[i, 6]), 3:6], N[int(N[i, 7]), 3:6])
        nested &= bool(np.array_equal(lo, N[i, :3]) and np.array_equal(hi, N[i, 3:6]))  # (tight: the union of the children)
    return total / area(0), depth, sorted(leaves), nested, seen


def test_reinsertion_keeps_a_valid_tight_tree_and_never_raises_the_fork_area():
    """glrt_bvh_reinsert (round 5): every triangle still in exactly one leaf, every fork's box the union of its children's, root = node 0 in DFS pre-order, the summed
    fork area (what it reports) not above the builder's and what this test measures itself; a second run finds little left; the oracle renders the same image through
    it; trees it has nothing to do with (fewer than four triangles, non-finite boxes) come back unchanged with depth -1; malformed ones are refused."""
    from oracle import pt_oracle
    for cfg, kw in (("c2", dict(width=40, height=24, subdiv=1)), ("c5", dict(width=24, height=16, n=4000)), ("c1", dict(width=32, height=32, subdiv=1))):
        sc, pr = scenes.CONFIGS[cfg](**kw)
        raw = np.asarray(sc["bvh_builder"], np.float32).reshape(-1, 3)
        out, depth, moved, cost = host.reinsert(raw)
        c0, _, leaves0, ok0, n0 = _tree_facts(raw)
        c1, d1, leaves1, ok1, n1 = _tree_facts(out)
        assert ok0 and ok1 and leaves0 == leaves1 and n0 == n1 == raw.shape[0] // 3
        assert d1 == depth and moved > 0
        assert abs(c0 - cost[0]) <= 1e-6 * c0 and abs(c1 - cost[1]) <= 1e-6 * c1 and c1 < c0
        N = out.reshape(-1, 9)
        forks = np.nonzero(N[:, 8] < 0)[0]
        assert (N[forks, 6] == forks + 1).all(), "DFS pre-order: a fork's x child follows it"
        again, _, moved2, cost2 = host.reinsert(out)
        assert cost2[1] <= cost2[0] * (1 + 1e-12) and cost2[1] > 0.98 * cost[1]
        ordered, _ = host.lights_first(out, sc["tri"], sc["mat"])
        a, ra = pt_oracle.render(dict(sc, bvh=ordered), pr)
        b, rb = pt_oracle.render(sc, pr)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and ra == rb
    # left alone
    sc, _ = scenes.config_c5(8, 8, n=3)
    raw = np.asarray(sc["bvh_builder"], np.float32).reshape(-1, 3)
    out, depth, moved, cost = host.reinsert(raw)
    assert depth == -1 and moved == 0 and np.array_equal(out, raw)
    sc, _ = scenes.config_c5(8, 8, n=200)
    bad = np.asarray(sc["bvh_builder"], np.float32).reshape(-1, 9).copy()
    bad[5, 3] = np.inf
    out, depth, moved, _ = host.reinsert(bad.reshape(-1, 3))
    assert depth == -1 and moved == 0 and np.array_equal(out.view(np.uint32), bad.reshape(-1, 3).view(np.uint32))
    # malformed: a child index out of range, a node referenced twice
    for col, val in ((6, 1e6), (6, None)):
        t = np.asarray(sc["bvh_builder"], np.float32).reshape(-1, 9).copy()
        t[0, col] = t[0, 7] if val is None else val
        with pytest.raises(RuntimeError):
            host.reinsert(t.reshape(-1, 3))


def test_the_sah_reinsert_builder_is_sah_plus_the_pass():
    sc, _ = scenes.config_c2(16, 16, subdiv=1)
    want, depth, _, _ = host.reinsert(host.build_bvh(sc["vert"], sc["tri"], "sah")[0])
    got = scenes.rebuild_bvh(sc, "sah-reinsert")
    assert np.array_equal(np.asarray(got["bvh_builder"]).view(np.uint32), want.view(np.uint32)) and got["bvh_depth"] == depth
    assert np.array_equal(np.asarray(got["bvh"]), host.lights_first(want, sc["tri"], sc["mat"])[0])


def test_lights_first_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("GLRT_BVH_LIGHTS_FIRST", "0")
    sc, _ = scenes.config_c2(16, 16, subdiv=1)
    assert sc["bvh_lights_first"] == 0 and np.array_equal(np.asarray(sc["bvh"]), np.asarray(sc["bvh_builder"]))


@pytest.mark.parametrize("exp", ["0", None, "1"])
def test_order_by_hits_puts_the_busier_child_first(monkeypatch, exp):
    """glrt_bvh_order_by_hits: at every fork the child whose subtree collected more hits PER UNIT COST (a subtree of n triangles is charged n^e, e = 0.5 by default,
    GLRT_HITS_COST_EXP; e = 0: hits alone) goes into children.y (the slot raytrace.frag:299-307 pops first); equal scores keep the builder's order; boxes, leaves and the
    set of triangles are untouched.  glrt_bvh_add_shadow_hits adds (all hits) / (light triangles) to every emitter."""
    if exp is None:
        monkeypatch.delenv("GLRT_HITS_COST_EXP", raising=False)
    else:
        monkeypatch.setenv("GLRT_HITS_COST_EXP", exp)
    e = 0.5 if exp is None else float(exp)
    sc, _ = scenes.config_c1(32, 32, subdiv=1)
    nodes = np.asarray(sc["bvh"], np.float32).reshape(-1, 9)
    n_tri = sc["tri"].shape[0]
    rng = np.random.default_rng(3)
    hits = rng.integers(0, 50, n_tri).astype(np.uint32)
    out, exchanged = host.order_by_hits(sc["bvh"], hits)
    out = out.reshape(-1, 9)
    assert exchanged > 0 and np.array_equal(out[:, :6], nodes[:, :6]) and np.array_equal(out[:, 8], nodes[:, 8])

    def subtree_hits(n, i):
        return int(hits[int(n[i, 8])]) if n[i, 8] >= 0 else sum(subtree_hits(n, int(c)) for c in n[i, 6:8] if c >= 0)

    def subtree_leaves(n, i):
        return 1 if n[i, 8] >= 0 else sum(subtree_leaves(n, int(c)) for c in n[i, 6:8] if c >= 0)
    for i in range(out.shape[0]):
        if out[i, 8] < 0 and out[i, 6] >= 0 and out[i, 7] >= 0:
            hx, hy = subtree_hits(out, int(out[i, 6])), subtree_hits(out, int(out[i, 7]))
            cx, cy = subtree_leaves(out, int(out[i, 6])), subtree_leaves(out, int(out[i, 7]))
            sx, sy = hx * float(cy) ** e, hy * float(cx) ** e  # hx / cx^e against hy / cy^e, cross-multiplied as the library does
            assert sy >= sx * (1 - 1e-12)
            assert {out[i, 6], out[i, 7]} == {nodes[i, 6], nodes[i, 7]}
            if hx == hy and cx == cy:
                assert out[i, 6] == nodes[i, 6]
    same, n0 = host.order_by_hits(sc["bvh"], np.zeros(n_tri, np.uint32))
    assert n0 == 0 and np.array_equal(same.reshape(-1, 9), nodes)
    # the shadow rays' share
    h2 = hits.copy()
    L = host.lib()
    import ctypes as C
    tri, mat = np.ascontiguousarray(sc["tri"], np.float32).reshape(-1, 4), np.ascontiguousarray(sc["mat"], np.float32).reshape(-1, 18)
    n_light = L.glrt_bvh_add_shadow_hits(h2.ctypes.data_as(C.POINTER(C.c_uint32)), n_tri, tri.ctypes.data_as(C.POINTER(C.c_float)), mat.ctypes.data_as(C.POINTER(C.c_float)), mat.shape[0])
    assert n_light == sc["light"].shape[0] > 0
    emit = np.array([np.linalg.norm(mat[int(t[3]), 3:6]) != 0 for t in tri])
    assert np.array_equal(h2[~emit], hits[~emit]) and np.all(h2[emit] == hits[emit] + int(hits.sum()) // n_light)


def test_sah_count_weight_switch_builds_another_valid_tree():
    """glrt_bvh_build_sah charges a subtree of n triangles n^0.8 in its split cost (host/bvh.cpp: weight(), profiles/r06_sah_count_weight.txt); GLRT_SAH_ALPHA=1 restores the
    plain surface-area heuristic.  The switch is read once per process, so the plain tree is built in a child: both are valid trees over all triangles within the depth
    bound, they differ, and the plain one is the better tree by the plain heuristic's own measure (the sum of its forks' areas weighted by triangle counts)."""
    import pickle
    import subprocess
    import sys
    code = ("import sys, pickle; sys.path[:0] = %r; from glrt_amd import scenes; import numpy as np; "
            "sc, _ = scenes.config_c2(16, 16, bvh='sah', subdiv=2); pickle.dump((np.asarray(sc['bvh_builder']), sc['bvh_depth']), sys.stdout.buffer)") % (sys.path[:3],)
    import os
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, env=dict(os.environ, GLRT_SAH_ALPHA="1"), timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    plain, plain_depth = pickle.loads(r.stdout)
    sc, _ = scenes.config_c2(16, 16, bvh="sah", subdiv=2)
    n_tri = sc["tri"].shape[0]
    mine = np.asarray(sc["bvh_builder"], np.float32)
    _tree_ok(mine, n_tri); _tree_ok(plain, n_tri)
    assert sc["bvh_depth"] < 63 and plain_depth < 63
    assert not np.array_equal(mine.reshape(-1, 9), np.asarray(plain, np.float32).reshape(-1, 9))

    def linear_cost(nodes):
        N = np.asarray(nodes, np.float32).reshape(-1, 9)
        cnt = np.zeros(N.shape[0], np.int64)
        order, st = [], [0]
        while st:
            i = st.pop(); order.append(i)
            if N[i, 8] < 0:
                st += [int(N[i, 6]), int(N[i, 7])]
        for i in reversed(order):
            cnt[i] = 1 if N[i, 8] >= 0 else cnt[int(N[i, 6])] + cnt[int(N[i, 7])]
        d = (N[:, 3:6] - N[:, 0:3]).astype(np.float64)
        area = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]
        return float((area * cnt)[N[:, 8] < 0].sum())
    assert linear_cost(plain) <= linear_cost(mine)
