// bvh.cpp -- host-side BVH construction for the flat 'u_bvhBuffer' wire format.
//
// Replaces the role of the reference's BVH::construct / constructRec
// (src/core/bvh.cpp:59-160) with an independent design: a full three-axis
// binned-SAH builder (the reference bins the longest centroid axis only), an
// explicit depth cap so the traversal stack bound is known at upload time, and
// a degenerate "chain" builder used to express brute-force (no-BVH) scans in
// the same node format.
//
// Wire format produced (bvh.h:84-100 of the reference):
//   node = 3 x vec3 = { bboxMin, bboxMax, children }
//   fork : children = (left, right, -1)      leaf : children = (-1, -1, triIndex)
//   root = node 0, nodes in DFS pre-order, exactly one triangle per leaf.
// A closest hit does not depend on the tree shape except at exact ties and where the reference's float slab test
// rejects a box whose triangle the ray would hit (grazing rays, flat boxes: a few pixels in ten thousand): the
// device layer is bit-exact for the tree it is handed, whichever builder made it.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>
#if defined(__SSE__)
#include <xmmintrin.h>
#endif

#include "glrt_host.h"

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::numeric_limits<float>::infinity();
            hi[a] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const float *p) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void grow(const Box &b) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0.f) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim {
    Box box;
    float c[3];
    int tri;
};

// A box's centre as the builders order and bin by it: 0 where it is not finite (a vertex at infinity, or at +inf and -inf at once: NaN), so that every
// comparison below is a total order and every float -> int conversion is in range.  The boxes themselves keep their values.  (csrc/lbvh.hip.h: the same.)
inline float centre(float lo, float hi) {
    const float c = 0.5f * (lo + hi);
    return (c - c == 0.0f) ? c : 0.0f;
}
// Bin of a centre on an axis of the centroid box; anything that is not a number in range (an extent that overflowed: scale = 0, inf * 0) goes to bin 0.
inline int bin_of(float c, float lo, float scale, int bins) {
    const float f = (c - lo) * scale;
    if (!(f >= 0.0f)) return 0;
    return f >= (float)bins ? bins - 1 : (int)f;
}

constexpr int kBins = 16;
// Past this depth the builder switches to median splits, which bounds the total
// depth by kSahDepthCap + ceil(log2(n)) (< 64 for n < 2^23: the reference
// shader's stack is int[64], raytrace.frag:284).
constexpr int kSahDepthCap = 40;

// What a subtree of n triangles is charged in the split cost (round 6, last session).  The surface-area heuristic charges n -- a leaf of n triangles tests them all --, but
// every subtree here is built down to ONE triangle per leaf (the wire format), where a subtree of n triangles costs a ray that enters it far less than n: charging
// n^0.8 makes the builder give up some balance for smaller boxes near the top.  Measured inside one context, trees alternating, images bit-identical
// (profiles/r06_sah_count_weight.txt): headline -0.6 ... -0.8 %, config 2 -0.8 %, config 4 -0.6 ... -2.5 %, config 5 -0.1 ... -0.3 %; every exponent from 0.65 to 0.85
// within 0.2 % of the others, 0.5 and below lose it again to depth (0.35: +5 % / +25 %), 1 + log2 n: +26 % / +119 %.  GLRT_SAH_ALPHA=1 restores the plain heuristic (A/B).
inline float weight(int n) {
    static const float alpha = [] { const char *e = std::getenv("GLRT_SAH_ALPHA"); const float a = e ? (float)std::atof(e) : 0.8f; return a > 0.0f && a <= 1.0f ? a : 0.8f; }();
    return alpha == 1.0f ? (float)n : std::pow((float)n, alpha);
}

struct Builder {
    std::vector<Prim> prims;
    float *nodes;  // 9 floats per node
    int next = 0;
    int max_depth = 0;

    void write(int idx, const Box &b, float cx, float cy, float cz) {
        float *n = nodes + 9 * (size_t)idx;
        n[0] = b.lo[0]; n[1] = b.lo[1]; n[2] = b.lo[2];
        n[3] = b.hi[0]; n[4] = b.hi[1]; n[5] = b.hi[2];
        n[6] = cx; n[7] = cy; n[8] = cz;
    }

    int build(int l, int r, int depth) {
        const int idx = next++;
        max_depth = std::max(max_depth, depth);
        Box bounds, cb;
        bounds.reset();
        cb.reset();
        for (int i = l; i < r; i++) {
            bounds.grow(prims[i].box);
            cb.grow(prims[i].c);
        }
        if (r - l == 1) {
            write(idx, bounds, -1.f, -1.f, (float)prims[l].tri);
            return idx;
        }
        int mid = -1;
        if (depth < kSahDepthCap && r - l > 2) {
            float best = std::numeric_limits<float>::infinity();
            int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; a++) {
                const float ext = cb.hi[a] - cb.lo[a];
                if (!(ext > 0.f)) continue;
                Box bb[kBins];
                int cnt[kBins] = {0};
                for (auto &b : bb) b.reset();
                const float scale = (float)kBins / ext;
                for (int i = l; i < r; i++) {
                    const int k = bin_of(prims[i].c[a], cb.lo[a], scale, kBins);
                    cnt[k]++;
                    bb[k].grow(prims[i].box);
                }
                float right_area[kBins];
                int right_cnt[kBins];
                Box acc;
                acc.reset();
                int c = 0;
                for (int k = kBins - 1; k > 0; k--) {
                    acc.grow(bb[k]);
                    c += cnt[k];
                    right_area[k] = acc.half_area();
                    right_cnt[k] = c;
                }
                acc.reset();
                c = 0;
                for (int k = 0; k < kBins - 1; k++) {
                    acc.grow(bb[k]);
                    c += cnt[k];
                    if (c == 0 || right_cnt[k + 1] == 0) continue;
                    float cost = acc.half_area() * weight(c) + right_area[k + 1] * weight(right_cnt[k + 1]);
                    if (cost < best) {
                        best = cost;
                        best_axis = a;
                        best_bin = k;
                    }
                }
            }
            if (best_axis >= 0) {
                const int a = best_axis;
                const float scale = (float)kBins / (cb.hi[a] - cb.lo[a]);
                const float lo = cb.lo[a];
                auto it = std::partition(prims.begin() + l, prims.begin() + r, [&](const Prim &p) {
                    return bin_of(p.c[a], lo, scale, kBins) <= best_bin;
                });
                mid = (int)(it - prims.begin());
                if (mid == l || mid == r) mid = -1;
            }
        }
        if (mid < 0) {
            int a = 0;
            for (int k = 1; k < 3; k++)
                if (cb.hi[k] - cb.lo[k] > cb.hi[a] - cb.lo[a]) a = k;
            mid = (l + r) / 2;
            std::nth_element(prims.begin() + l, prims.begin() + mid, prims.begin() + r,
                             [a](const Prim &x, const Prim &y) { return x.c[a] < y.c[a]; });
        }
        const int left = build(l, mid, depth + 1);
        const int right = build(mid, r, depth + 1);
        write(idx, bounds, (float)left, (float)right, -1.f);
        return idx;
    }
};

bool load_prims(const float *vert, size_t n_vert, const float *tri, size_t n_tri, std::vector<Prim> &out) {
    out.resize(n_tri);
    for (size_t t = 0; t < n_tri; t++) {
        Prim &p = out[t];
        p.box.reset();
        p.tri = (int)t;
        for (int k = 0; k < 3; k++) {
            const float fi = tri[4 * t + k];
            if (!(fi >= 0.f) || (size_t)fi >= n_vert) return false;
            p.box.grow(vert + GLRT_VERTEX_FLOATS * (size_t)fi);
        }
        for (int a = 0; a < 3; a++) p.c[a] = centre(p.box.lo[a], p.box.hi[a]);
    }
    return true;
}

}  // namespace

extern "C" {

size_t glrt_bvh_node_count(size_t n_tri) { return n_tri ? 2 * n_tri - 1 : 0; }

int glrt_bvh_build_sah(const float *vert, size_t n_vert, const float *tri, size_t n_tri, float *nodes_out,
                       int *max_depth_out) {
    if (!vert || !tri || !nodes_out || n_tri == 0) return GLRT_HOST_EINVAL;
    Builder b;
    if (!load_prims(vert, n_vert, tri, n_tri, b.prims)) return GLRT_HOST_EINDEX;
    b.nodes = nodes_out;
    b.build(0, (int)n_tri, 0);
    if (max_depth_out) *max_depth_out = b.max_depth;
    return b.max_depth < 63 ? GLRT_HOST_OK : GLRT_HOST_EDEPTH;
}

// ---------------------------------------------------------------------------------------------- LBVH
// Linear BVH (Karras 2012): 30-bit Morton codes of the triangle-box centres, made unique by appending the triangle
// index, sorted; the hierarchy follows from the common-prefix lengths of neighbouring keys.  This is the CPU statement
// of the algorithm; the GPU builder (csrc/lbvh.hip, glrtx_build_lbvh) produces the same nodes bit for bit -- every
// step is either integer arithmetic or an exactly rounded float operation, and box unions are exact.
// Output layout: internal node i at index i (root = 0), the leaf of sorted position k at index (n - 1) + k.
// The Morton tree is then improved by GLRT_LBVH_ROTATION_PASSES sweeps of tree rotations (rotate_tree below) and a rebuild of
// its small subtrees with the exact sweep SAH (rebuild_subtrees below).
namespace lbvh {

inline uint32_t expand10(uint32_t v) {  // 10 bits -> every third bit
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

inline uint32_t quantize(float c, float lo, float ext) {
    if (!(ext > 0.0f)) return 0u;
    const float q = (c - lo) / ext * 1024.0f;
    if (!(q >= 0.0f)) return 0u;  // (also a NaN: an extent that overflowed)
    return q >= 1024.0f ? 1023u : (uint32_t)(int)q;
}

inline int delta(const std::vector<uint64_t> &k, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __builtin_clzll(k[(size_t)i] ^ k[(size_t)j]);  // keys are unique: the xor is never 0
}

// ---- Tree rotations (Kensler 2008), the quality pass that follows the Morton build.  A Morton tree splits at the spatial
// median of a fixed axis cycle; its surface-area cost is 8-28 % above the binned-SAH tree's on the BASELINE scenes and it renders
// 9-15 % slower.  One rotation at node i exchanges one child of i with a grandchild on the other side, or a grandchild of one
// child with a grandchild of the other, when that shrinks the summed surface area of i's children (six candidates, the best
// strictly negative change wins, the first in the order below on ties); the box of i itself never changes.  A sweep goes bottom-up by the depth the nodes have when
// the sweep STARTS: nodes of equal depth have disjoint subtrees, a rotation only rearranges the subtree of its own node, and
// within a sweep nothing above the current level has moved yet, so a level's nodes can be processed in any order -- or, on
// the device, in parallel (csrc/lbvh.hip.h: k_rotate_level) -- with the same result.  Float arithmetic: differences, products and sums of box extents in one fixed association, exact min/max.
// Returns the depth of the deepest leaf of the rotated tree.
inline float half_area9(const float *lo, const float *hi) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return (dx * dy + dy * dz) + dz * dx;
}
inline float union_area9(const float *a, const float *b) {  // a, b: node records {lo[3], hi[3], ...}
    float lo[3], hi[3];
    for (int k = 0; k < 3; k++) { lo[k] = std::min(a[k], b[k]); hi[k] = std::max(a[3 + k], b[3 + k]); }
    return half_area9(lo, hi);
}
inline void rotate_node(float *nodes, int n, int i) {
    float *N = nodes + 9 * (size_t)i;
    const int c[2] = {(int)N[6], (int)N[7]};
    float best = 0.0f;
    int kind = -1, bside = 0, bwhich = 0;  // kind 0: child <-> grandchild, 1: grandchild <-> grandchild
    for (int side = 0; side < 2; side++) {  // x = the child that is restructured, y = its sibling
        const int x = c[side], y = c[side ^ 1];
        if (x >= n - 1) continue;  // a leaf has no grandchildren to give
        const float *X = nodes + 9 * (size_t)x, *Y = nodes + 9 * (size_t)y;
        const float old = half_area9(X, X + 3);
        const float g0 = union_area9(Y, nodes + 9 * (size_t)(int)X[7]) - old;  // y <-> x.left : x' = (y, x.right)
        const float g1 = union_area9(nodes + 9 * (size_t)(int)X[6], Y) - old;  // y <-> x.right: x' = (x.left, y)
        if (g0 < best) { best = g0; kind = 0; bside = side; bwhich = 0; }
        if (g1 < best) { best = g1; kind = 0; bside = side; bwhich = 1; }
    }
    if (c[0] < n - 1 && c[1] < n - 1) {  // both children internal: exchange a grandchild of one with a grandchild of the other
        const float *A = nodes + 9 * (size_t)c[0], *B = nodes + 9 * (size_t)c[1];
        const float *A1 = nodes + 9 * (size_t)(int)A[6], *A2 = nodes + 9 * (size_t)(int)A[7];
        const float *B1 = nodes + 9 * (size_t)(int)B[6], *B2 = nodes + 9 * (size_t)(int)B[7];
        const float old = half_area9(A, A + 3) + half_area9(B, B + 3);
        const float h0 = (union_area9(B1, A2) + union_area9(A1, B2)) - old;  // a.left <-> b.left : a' = (b1, a2), b' = (a1, b2)
        const float h1 = (union_area9(B2, A2) + union_area9(B1, A1)) - old;  // a.left <-> b.right: a' = (b2, a2), b' = (b1, a1)
        if (h0 < best) { best = h0; kind = 1; bwhich = 0; }
        if (h1 < best) { best = h1; kind = 1; bwhich = 1; }
    }
    if (kind < 0) return;
    auto refit = [&](float *X) {
        const float *P = nodes + 9 * (size_t)(int)X[6], *Q = nodes + 9 * (size_t)(int)X[7];
        for (int k = 0; k < 3; k++) { X[k] = std::min(P[k], Q[k]); X[3 + k] = std::max(P[3 + k], Q[3 + k]); }
    };
    if (kind == 0) {
        float *X = nodes + 9 * (size_t)c[bside];
        const int y = c[bside ^ 1];
        const int moved = (int)X[6 + bwhich];
        X[6 + bwhich] = (float)y;
        N[6 + (bside ^ 1)] = (float)moved;
        refit(X);
    } else {
        float *A = nodes + 9 * (size_t)c[0], *B = nodes + 9 * (size_t)c[1];
        const float a1 = A[6];
        A[6] = B[6 + bwhich];
        B[6 + bwhich] = a1;
        refit(A);
        refit(B);
    }
}
inline int rotate_tree(float *nodes, int n, int passes) {
    if (n < 3) return n - 1;  // 1 triangle: depth 0; 2: depth 1; nothing to rotate
    std::vector<int> level((size_t)n - 1, 0), order;
    order.reserve((size_t)n - 1);
    for (int p = 0; p < passes; p++) {
        // depths at the START of the sweep: a rotation at an ancestor moves whole subtrees between nodes of the same depth, so the
        // levels of the previous sweep would no longer be sets of nodes with disjoint subtrees
        order.assign(1, 0);
        level[0] = 0;
        int deepest = 0;
        for (size_t q = 0; q < order.size(); q++) {
            const int i = order[q];
            for (int k = 6; k <= 7; k++) {
                const int c = (int)nodes[9 * (size_t)i + k];
                if (c < n - 1) { level[(size_t)c] = level[(size_t)i] + 1; deepest = std::max(deepest, level[(size_t)c]); order.push_back(c); }
            }
        }
        std::vector<std::vector<int>> by_level((size_t)deepest + 1);
        for (int i : order) by_level[(size_t)level[(size_t)i]].push_back(i);
        for (int d = deepest; d >= 0; d--)
            for (int i : by_level[(size_t)d]) rotate_node(nodes, n, i);
    }
    // depth of the deepest leaf of the rotated tree
    int max_depth = 0;
    std::vector<std::pair<int, int>> st{{0, 0}};
    while (!st.empty()) {
        const auto [i, d] = st.back();
        st.pop_back();
        for (int k = 6; k <= 7; k++) {
            const int c = (int)nodes[9 * (size_t)i + k];
            max_depth = std::max(max_depth, d + 1);
            if (c < n - 1) st.push_back({c, d + 1});
        }
    }
    return max_depth;
}


// ---- Subtree rebuild, the second quality pass.  Rotations converge to a local optimum that leaves the bottom of a Morton tree
// 6 % behind a SAH tree in expected triangle tests (and config 5 5-6 % behind in render time).  Every MAXIMAL subtree with at
// most kRebuildLeaves leaves is therefore rebuilt from its leaves, top-down, with the exact sweep SAH: at every node all three axes,
// every split position of the leaves sorted by (box centre, leaf rank), cost = area(left) * count(left) + area(right) * count(right),
// among equal costs the most balanced split, then the first in (axis, position) order, wins (equal boxes would otherwise give
// a chain); after kRebuildSahLevels levels the remaining segments are halved along the first axis, which bounds the subtree's depth.  The subtree keeps its node slots -- its root stays where it is, the other
// internal slots are handed out in ascending index order, breadth-first (level by level, segments in position order, left child
// before right child) -- so nothing outside the subtree changes.  Everything is a total order or an exact min / max / single
// rounded float operation, so the device version (csrc/lbvh.hip.h: k_rebuild_subtrees, one wave per subtree, level-synchronous)
// produces the same nodes bit for bit.
struct TBox {  // unions of boxes without negative zeros (grow() reads x + 0.0f): min / max give the same box in any order of the operands
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -std::numeric_limits<float>::infinity(); } }
    void grow(const float *l, const float *h) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::fmin(lo[a], l[a] + 0.0f);
            hi[a] = std::fmax(hi[a], h[a] + 0.0f);
        }
    }
};
constexpr int kRebuildSahLevels = 20;
// One subtree rebuilt from its leaves with the exact sweep SAH: root slot r, the other internal slots and the leaf nodes in ascending index order.
inline void rebuild_one(float *nodes, int r, const std::vector<int> &slots, const std::vector<int> &leaves) {
    std::vector<int> ord, tmp, next_ord;
    struct Seg { int a, b, slot; };
    std::vector<Seg> segs, next;
    std::vector<float> cen;
        const int m = (int)leaves.size();
        cen.resize(3 * (size_t)m);
        for (int k = 0; k < m; k++) {
            const float *L = nodes + 9 * (size_t)leaves[(size_t)k];
            for (int a = 0; a < 3; a++) cen[3 * (size_t)k + a] = centre(L[a], L[3 + a]);
        }
        ord.resize((size_t)m);
        for (int k = 0; k < m; k++) ord[(size_t)k] = k;
        next_ord.resize((size_t)m);
        segs.assign(1, Seg{0, m, r});
        size_t next_slot = 0;
        for (int level = 0; !segs.empty(); level++) {
            next.clear();
            for (const Seg &S : segs) {
                const int ns = S.b - S.a;
                float best = std::numeric_limits<float>::infinity();
                int best_axis = 0, best_k = 1, best_imb = ns;
                TBox all;
                all.reset();
                for (int a = 0; a < 3; a++) {
                    tmp.assign(ord.begin() + S.a, ord.begin() + S.b);
                    std::sort(tmp.begin(), tmp.end(), [&](int x, int y) {
                        const float cx = cen[3 * (size_t)x + a], cy = cen[3 * (size_t)y + a];
                        return cx < cy || (cx == cy && x < y);
                    });
                    std::vector<TBox> suf((size_t)ns + 1);
                    suf[(size_t)ns].reset();
                    for (int j = ns - 1; j >= 0; j--) {
                        suf[(size_t)j] = suf[(size_t)j + 1];
                        const float *L = nodes + 9 * (size_t)leaves[(size_t)tmp[(size_t)j]];
                        suf[(size_t)j].grow(L, L + 3);
                    }
                    if (a == 0) all = suf[0];
                    TBox pre;
                    pre.reset();
                    for (int k = 1; k < ns; k++) {
                        const float *L = nodes + 9 * (size_t)leaves[(size_t)tmp[(size_t)k - 1]];
                        pre.grow(L, L + 3);
                        const float cost = half_area9(pre.lo, pre.hi) * (float)k + half_area9(suf[(size_t)k].lo, suf[(size_t)k].hi) * (float)(ns - k);
                        const int imb = std::abs(2 * k - ns);
                        if (level < kRebuildSahLevels && (cost < best || (cost == best && imb < best_imb))) { best = cost; best_axis = a; best_k = k; best_imb = imb; }
                    }
                    if (a == best_axis) std::copy(tmp.begin(), tmp.end(), next_ord.begin() + S.a);  // the order of the winning axis so far
                }
                std::copy(next_ord.begin() + S.a, next_ord.begin() + S.b, ord.begin() + S.a);
                if (level >= kRebuildSahLevels) best_k = ns / 2;
                const Seg L{S.a, S.a + best_k, 0}, R{S.a + best_k, S.b, 0};
                int refs[2];
                const Seg *ch[2] = {&L, &R};
                for (int k = 0; k < 2; k++) {
                    if (ch[k]->b - ch[k]->a == 1) refs[k] = leaves[(size_t)ord[(size_t)ch[k]->a]];
                    else {
                        refs[k] = slots[next_slot++];
                        next.push_back(Seg{ch[k]->a, ch[k]->b, refs[k]});
                    }
                }
                float *N = nodes + 9 * (size_t)S.slot;
                for (int k = 0; k < 3; k++) { N[k] = all.lo[k]; N[3 + k] = all.hi[k]; }
                N[6] = (float)refs[0]; N[7] = (float)refs[1]; N[8] = -1.0f;
            }
            segs.swap(next);
        }
}
inline void rebuild_subtrees(float *nodes, int n, int max_leaves) {
    if (n < 3 || max_leaves < 3) return;
    const int n_int = n - 1;
    // leaves below every internal node, capped at max_leaves + 1
    std::vector<int> count((size_t)n_int, 0), parent((size_t)n_int, -1), order;
    order.reserve((size_t)n_int);
    order.push_back(0);
    for (size_t q = 0; q < order.size(); q++) {
        const int i = order[q];
        for (int k = 6; k <= 7; k++) {
            const int c = (int)nodes[9 * (size_t)i + k];
            if (c < n_int) { parent[(size_t)c] = i; order.push_back(c); }
        }
    }
    for (size_t q = order.size(); q-- > 0;) {
        const int i = order[q];
        int s = 0;
        for (int k = 6; k <= 7; k++) {
            const int c = (int)nodes[9 * (size_t)i + k];
            s += c < n_int ? count[(size_t)c] : 1;
        }
        count[(size_t)i] = std::min(s, max_leaves + 1);
    }
    std::vector<int> slots, leaves, tmp;
    for (int r : order) {
        if (count[(size_t)r] > max_leaves || (parent[(size_t)r] >= 0 && count[(size_t)parent[(size_t)r]] <= max_leaves)) continue;
        // the subtree's internal slots (without r) and leaf nodes, ascending
        slots.clear(); leaves.clear();
        tmp.assign(1, r);
        while (!tmp.empty()) {
            const int i = tmp.back();
            tmp.pop_back();
            if (i >= n_int) { leaves.push_back(i); continue; }
            if (i != r) slots.push_back(i);
            tmp.push_back((int)nodes[9 * (size_t)i + 6]);
            tmp.push_back((int)nodes[9 * (size_t)i + 7]);
        }
        std::sort(slots.begin(), slots.end());
        std::sort(leaves.begin(), leaves.end());
        rebuild_one(nodes, r, slots, leaves);
    }
}

// depth of the deepest leaf
inline int tree_depth(const float *nodes, int n) {
    int max_depth = 0;
    std::vector<std::pair<int, int>> st{{0, 0}};
    if (n < 2) return 0;
    while (!st.empty()) {
        const auto [i, d] = st.back();
        st.pop_back();
        for (int k = 6; k <= 7; k++) {
            const int c = (int)nodes[9 * (size_t)i + k];
            max_depth = std::max(max_depth, d + 1);
            if (c < n - 1) st.push_back({c, d + 1});
        }
    }
    return max_depth;
}

}  // namespace lbvh

int glrt_bvh_build_lbvh(const float *vert, size_t n_vert, const float *tri, size_t n_tri, float *nodes_out,
                        int *max_depth_out) {
    if (!vert || !tri || !nodes_out || n_tri == 0) return GLRT_HOST_EINVAL;
    if (2 * n_tri - 1 > ((size_t)1 << 24)) return GLRT_HOST_EINVAL;  // node indices travel as floats
    // This function is the CPU statement of the device build (csrc/lbvh.hip.h), which runs with fp32 denormals flushed like the rest of the device code:
    // the same mode here, for the duration of the call.
    struct FlushDenormals {
#if defined(__SSE__)
        unsigned csr = _mm_getcsr();
        FlushDenormals() { _mm_setcsr(csr | 0x8040u); }
        ~FlushDenormals() { _mm_setcsr(csr); }
#endif
    } flush_denormals;
    (void)flush_denormals;
    std::vector<Prim> prims;
    if (!load_prims(vert, n_vert, tri, n_tri, prims)) return GLRT_HOST_EINDEX;
    const int n = (int)n_tri;
    auto put = [&](size_t idx, const Box &b, float cx, float cy, float cz) {
        float *o = nodes_out + 9 * idx;
        o[0] = b.lo[0]; o[1] = b.lo[1]; o[2] = b.lo[2];
        o[3] = b.hi[0]; o[4] = b.hi[1]; o[5] = b.hi[2];
        o[6] = cx; o[7] = cy; o[8] = cz;
    };
    if (n == 1) {
        put(0, prims[0].box, -1.f, -1.f, 0.f);
        if (max_depth_out) *max_depth_out = 0;
        return GLRT_HOST_OK;
    }
    Box cb;
    cb.reset();
    for (auto &p : prims) cb.grow(p.c);
    std::vector<uint64_t> keys((size_t)n);
    for (int t = 0; t < n; t++) {
        const Prim &p = prims[(size_t)t];
        const uint32_t m = (lbvh::expand10(lbvh::quantize(p.c[0], cb.lo[0], cb.hi[0] - cb.lo[0])) << 2) |
                           (lbvh::expand10(lbvh::quantize(p.c[1], cb.lo[1], cb.hi[1] - cb.lo[1])) << 1) |
                           lbvh::expand10(lbvh::quantize(p.c[2], cb.lo[2], cb.hi[2] - cb.lo[2]));
        keys[(size_t)t] = ((uint64_t)m << 32) | (uint32_t)t;
    }
    std::sort(keys.begin(), keys.end());

    std::vector<int> left((size_t)n - 1), right((size_t)n - 1);  // >= 0 internal, < 0 ~leaf position
    for (int i = 0; i < n - 1; i++) {
        const int d = lbvh::delta(keys, n, i, i + 1) - lbvh::delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
        const int dmin = lbvh::delta(keys, n, i, i - d);
        int lmax = 2;
        while (lbvh::delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (lbvh::delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
        const int j = i + l * d;
        const int dnode = lbvh::delta(keys, n, i, j);
        int s = 0, t = l;
        do {
            t = (t + 1) >> 1;
            if (lbvh::delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        } while (t > 1);
        const int g = i + s * d + (d < 0 ? -1 : 0);
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        left[(size_t)i] = (lo == g) ? ~g : g;
        right[(size_t)i] = (hi == g + 1) ? ~(g + 1) : g + 1;
    }
    // boxes bottom-up (post-order walk from the root), depth on the way down
    std::vector<Box> ibox((size_t)n - 1);
    std::vector<int> order, depth((size_t)n - 1, 0);
    order.reserve((size_t)n - 1);
    std::vector<int> st{0};
    int max_depth = 0;
    while (!st.empty()) {
        const int i = st.back();
        st.pop_back();
        order.push_back(i);
        for (int c : {left[(size_t)i], right[(size_t)i]}) {
            if (c >= 0) { depth[(size_t)c] = depth[(size_t)i] + 1; st.push_back(c); }
            max_depth = std::max(max_depth, depth[(size_t)i] + 1);
        }
    }
    auto leaf_box = [&](int pos) -> const Box & { return prims[(size_t)(uint32_t)keys[(size_t)pos]].box; };
    for (size_t k = order.size(); k-- > 0;) {
        const int i = order[k];
        Box b;
        b.reset();
        for (int c : {left[(size_t)i], right[(size_t)i]}) b.grow(c >= 0 ? ibox[(size_t)c] : leaf_box(~c));
        ibox[(size_t)i] = b;
    }
    auto node_index = [&](int c) { return c >= 0 ? (float)c : (float)((n - 1) + ~c); };
    for (int i = 0; i < n - 1; i++) put((size_t)i, ibox[(size_t)i], node_index(left[(size_t)i]), node_index(right[(size_t)i]), -1.f);
    for (int k = 0; k < n; k++) put((size_t)(n - 1 + k), leaf_box(k), -1.f, -1.f, (float)(uint32_t)keys[(size_t)k]);
    lbvh::rotate_tree(nodes_out, n, GLRT_LBVH_ROTATION_PASSES);
    lbvh::rebuild_subtrees(nodes_out, n, GLRT_LBVH_REBUILD_LEAVES);
    max_depth = lbvh::tree_depth(nodes_out, n);
    if (max_depth_out) *max_depth_out = max_depth;
    return max_depth < 63 ? GLRT_HOST_OK : GLRT_HOST_EDEPTH;
}

// ---------------------------------------------------------------------------------------------- SAH by levels
// A top-down binned-SAH build that a GPU can run level by level (csrc/sahl.hip.h: glrtx_build_bvh_sah is the device statement and produces the same nodes bit for
// bit), joined to the exact-sweep builder of the LBVH pass at the bottom.  Why it exists (round 5, profiles/r05_tree_study.txt): the Morton tree with rotations and
// 64-leaf rebuilds renders config 5 3.6 % slower than the CPU binned-SAH tree; rebuilding larger and larger subtrees closes the gap only at >= 8192 leaves -- it is the
// TOP of a Morton tree that is behind -- while the whole-tree exact sweep is no better than the binned tree (+-0.4 %).  So: the top by binned SAH, the bottom by exact sweep.
//   * triangles in Morton order (the LBVH's keys): position k of that order is leaf node (n - 1) + k, as in the LBVH layout; internal nodes are 0 .. n - 2, root 0;
//   * a SEGMENT is a set of positions; the root segment holds all of them.  Level by level every OPEN segment (more than kClosed = GLRT_LBVH_REBUILD_LEAVES members) is
//     split: bounds of its members' box centres; per axis with a positive extent 16 bins (bin_of), a count and a box per bin; the split with the lowest
//     area(left) * count(left) + area(right) * count(right) over the 3 x 15 bin boundaries with members on both sides, the first in (axis, bin) order among equals;
//     members with bin <= the chosen bin go left.  If no boundary qualifies (all centres in one bin on every axis) or the segment is kDepthCap levels deep, the segment is
//     split by POSITION instead: 16 bins over its range of positions, the boundary with the most even counts (first among equals) -- positions are distinct, so this
//     always separates something, and it bounds the depth;
//   * a child with one member is that leaf; with 2 .. kClosed members it is CLOSED: it gets a root node now and is built afterwards by the exact sweep SAH of the LBVH
//     pass (lbvh::rebuild_one) from its leaves; a larger child is open in the next level.  Node numbers: the internal children of a level in segment order, left before
//     right, continue the running count (breadth-first); the closed subtrees' other nodes follow behind all of them, subtree after subtree in the order they were closed;
//   * boxes: a closed subtree's come from rebuild_one; the nodes above are the unions of their children's boxes, read as x + 0.0f (no negative zeros: min / max are then
//     independent of the order of the operands, which is what lets the device form the same unions with atomics).
namespace sahl {
constexpr int kClosed = GLRT_LBVH_REBUILD_LEAVES;
constexpr int kDepthCap = 40;
struct BinSet {
    int cnt[3][kBins];
    lbvh::TBox box[3][kBins];
    int rcnt[kBins];
};
}  // namespace sahl

int glrt_bvh_build_sah_levels(const float *vert, size_t n_vert, const float *tri, size_t n_tri, float *nodes_out, int *max_depth_out) {
    if (!vert || !tri || !nodes_out || n_tri == 0) return GLRT_HOST_EINVAL;
    if (2 * n_tri - 1 > ((size_t)1 << 24)) return GLRT_HOST_EINVAL;  // node indices travel as floats
    struct FlushDenormals {  // the device statement runs with fp32 denormals flushed (as glrt_bvh_build_lbvh)
#if defined(__SSE__)
        unsigned csr = _mm_getcsr();
        FlushDenormals() { _mm_setcsr(csr | 0x8040u); }
        ~FlushDenormals() { _mm_setcsr(csr); }
#endif
    } flush_denormals;
    (void)flush_denormals;
    std::vector<Prim> prims;
    if (!load_prims(vert, n_vert, tri, n_tri, prims)) return GLRT_HOST_EINDEX;
    const int n = (int)n_tri;
    auto put = [&](size_t idx, const float *lo, const float *hi, float cx, float cy, float cz) {
        float *o = nodes_out + 9 * idx;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2];
        o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
        o[6] = cx; o[7] = cy; o[8] = cz;
    };
    if (n == 1) {
        put(0, prims[0].box.lo, prims[0].box.hi, -1.f, -1.f, 0.f);
        if (max_depth_out) *max_depth_out = 0;
        return GLRT_HOST_OK;
    }
    // Morton order, exactly as glrt_bvh_build_lbvh
    Box cb0;
    cb0.reset();
    for (auto &p : prims) cb0.grow(p.c);
    std::vector<uint64_t> keys((size_t)n);
    for (int t = 0; t < n; t++) {
        const Prim &p = prims[(size_t)t];
        const uint32_t m = (lbvh::expand10(lbvh::quantize(p.c[0], cb0.lo[0], cb0.hi[0] - cb0.lo[0])) << 2) |
                           (lbvh::expand10(lbvh::quantize(p.c[1], cb0.lo[1], cb0.hi[1] - cb0.lo[1])) << 1) |
                           lbvh::expand10(lbvh::quantize(p.c[2], cb0.lo[2], cb0.hi[2] - cb0.lo[2]));
        keys[(size_t)t] = ((uint64_t)m << 32) | (uint32_t)t;
    }
    std::sort(keys.begin(), keys.end());
    const int n_int = n - 1;
    auto prim_at = [&](int pos) -> const Prim & { return prims[(size_t)(uint32_t)keys[(size_t)pos]]; };
    for (int k = 0; k < n; k++) put((size_t)(n_int + k), prim_at(k).box.lo, prim_at(k).box.hi, -1.f, -1.f, (float)(uint32_t)keys[(size_t)k]);

    using sahl::kClosed;
    // seg[pos]: index into the current level's open list, or -1 once the position's segment is closed / a leaf
    std::vector<int> seg((size_t)n, 0), closed_of((size_t)n, -1);
    struct Open { int node, count, depth; };
    std::vector<Open> open, next_open;
    struct Closed { int root, count; };
    std::vector<Closed> closed;
    int next_id = 1;  // node 0 is the root segment's node
    if (n <= kClosed) {
        closed.push_back(Closed{0, n});
        for (int k = 0; k < n; k++) { seg[(size_t)k] = -1; closed_of[(size_t)k] = 0; }
    } else {
        open.push_back(Open{0, n, 0});
    }
    std::vector<float> cbl, cbh;          // per open segment: bounds of the centres, 3 + 3
    std::vector<int> pmin, pmax;          // ... and of the positions
    std::vector<sahl::BinSet> bins;
    struct Split { int axis, bin, nl, nr; float lo, scale; int p0, prange; };  // axis 3: by position
    std::vector<Split> split;
    const float inf = std::numeric_limits<float>::infinity();
    while (!open.empty()) {
        const size_t S = open.size();
        cbl.assign(3 * S, inf); cbh.assign(3 * S, -inf);
        pmin.assign(S, n); pmax.assign(S, -1);
        for (int k = 0; k < n; k++) {
            const int s_ = seg[(size_t)k];
            if (s_ < 0) continue;
            const Prim &p = prim_at(k);
            for (int a = 0; a < 3; a++) {
                cbl[3 * (size_t)s_ + a] = std::fmin(cbl[3 * (size_t)s_ + a], p.c[a] + 0.0f);
                cbh[3 * (size_t)s_ + a] = std::fmax(cbh[3 * (size_t)s_ + a], p.c[a] + 0.0f);
            }
            pmin[(size_t)s_] = std::min(pmin[(size_t)s_], k);
            pmax[(size_t)s_] = std::max(pmax[(size_t)s_], k);
        }
        bins.resize(S);
        for (auto &b : bins) {
            for (int a = 0; a < 3; a++)
                for (int q = 0; q < kBins; q++) { b.cnt[a][q] = 0; b.box[a][q].reset(); }
            for (int q = 0; q < kBins; q++) b.rcnt[q] = 0;
        }
        auto rank_bin = [&](int k, int p0, int prange) { return (int)(((uint64_t)(uint32_t)(k - p0) * (uint64_t)kBins) / (uint64_t)(uint32_t)prange); };
        for (int k = 0; k < n; k++) {
            const int s_ = seg[(size_t)k];
            if (s_ < 0) continue;
            const Prim &p = prim_at(k);
            sahl::BinSet &b = bins[(size_t)s_];
            for (int a = 0; a < 3; a++) {
                const float lo = cbl[3 * (size_t)s_ + a], ext = cbh[3 * (size_t)s_ + a] - lo;
                if (!(ext > 0.f)) continue;
                const int q = bin_of(p.c[a], lo, (float)kBins / ext, kBins);
                b.cnt[a][q]++;
                b.box[a][q].grow(p.box.lo, p.box.hi);
            }
            b.rcnt[rank_bin(k, pmin[(size_t)s_], pmax[(size_t)s_] - pmin[(size_t)s_] + 1)]++;
        }
        split.resize(S);
        next_open.clear();
        // every open segment's split; children numbered in segment order, left before right
        std::vector<int> child_ref(2 * S), child_open(2 * S, -1), child_closed(2 * S, -1);
        for (size_t s_ = 0; s_ < S; s_++) {
            const sahl::BinSet &b = bins[s_];
            float best = inf;
            int best_axis = -1, best_bin = -1, best_nl = 0;
            if (open[s_].depth < sahl::kDepthCap)
                for (int a = 0; a < 3; a++) {
                    const float ext = cbh[3 * s_ + a] - cbl[3 * s_ + a];
                    if (!(ext > 0.f)) continue;
                    float right_area[kBins];
                    int right_cnt[kBins];
                    lbvh::TBox acc;
                    acc.reset();
                    int c = 0;
                    for (int q = kBins - 1; q > 0; q--) {
                        if (b.cnt[a][q]) acc.grow(b.box[a][q].lo, b.box[a][q].hi);
                        c += b.cnt[a][q];
                        right_area[q] = c ? lbvh::half_area9(acc.lo, acc.hi) : 0.f;
                        right_cnt[q] = c;
                    }
                    acc.reset();
                    c = 0;
                    for (int q = 0; q < kBins - 1; q++) {
                        if (b.cnt[a][q]) acc.grow(b.box[a][q].lo, b.box[a][q].hi);
                        c += b.cnt[a][q];
                        if (c == 0 || right_cnt[q + 1] == 0) continue;
                        const float cost = lbvh::half_area9(acc.lo, acc.hi) * (float)c + right_area[q + 1] * (float)right_cnt[q + 1];
                        if (cost < best) { best = cost; best_axis = a; best_bin = q; best_nl = c; }
                    }
                }
            Split &sp = split[s_];
            sp.p0 = pmin[s_]; sp.prange = pmax[s_] - pmin[s_] + 1;
            if (best_axis >= 0) {
                sp.axis = best_axis; sp.bin = best_bin; sp.nl = best_nl; sp.nr = open[s_].count - best_nl;
                sp.lo = cbl[3 * s_ + best_axis]; sp.scale = (float)kBins / (cbh[3 * s_ + best_axis] - cbl[3 * s_ + best_axis]);
            } else {  // by position: the most even boundary, the first among equals
                int c = 0, bb = -1, bnl = 0, bimb = 0;
                for (int q = 0; q < kBins - 1; q++) {
                    c += b.rcnt[q];
                    if (c == 0 || c == open[s_].count) continue;
                    const int imb = std::abs(2 * c - open[s_].count);
                    if (bb < 0 || imb < bimb) { bb = q; bnl = c; bimb = imb; }
                }
                sp.axis = 3; sp.bin = bb; sp.nl = bnl; sp.nr = open[s_].count - bnl; sp.lo = 0.f; sp.scale = 0.f;
            }
            const int cnts[2] = {sp.nl, sp.nr};
            for (int side = 0; side < 2; side++) {
                const int c = cnts[side];
                if (c == 1) { child_ref[2 * s_ + side] = -1; continue; }  // the member itself writes the leaf's number
                const int id = next_id++;
                child_ref[2 * s_ + side] = id;
                if (c > kClosed) { child_open[2 * s_ + side] = (int)next_open.size(); next_open.push_back(Open{id, c, open[s_].depth + 1}); }
                else { child_closed[2 * s_ + side] = (int)closed.size(); closed.push_back(Closed{id, c}); }
            }
            float *N = nodes_out + 9 * (size_t)open[s_].node;
            N[6] = (float)child_ref[2 * s_]; N[7] = (float)child_ref[2 * s_ + 1]; N[8] = -1.0f;
        }
        for (int k = 0; k < n; k++) {
            const int s_ = seg[(size_t)k];
            if (s_ < 0) continue;
            const Split &sp = split[(size_t)s_];
            const int q = sp.axis == 3 ? rank_bin(k, sp.p0, sp.prange) : bin_of(prim_at(k).c[sp.axis], sp.lo, sp.scale, kBins);
            const int side = q <= sp.bin ? 0 : 1;
            const size_t ci = 2 * (size_t)s_ + (size_t)side;
            if ((side ? sp.nr : sp.nl) == 1) {
                nodes_out[9 * (size_t)open[(size_t)s_].node + 6 + (size_t)side] = (float)(n_int + k);
                seg[(size_t)k] = -1;
            } else if (child_open[ci] >= 0) seg[(size_t)k] = child_open[ci];
            else { seg[(size_t)k] = -1; closed_of[(size_t)k] = child_closed[ci]; }
        }
        open.swap(next_open);
    }
    // the closed subtrees: their other nodes behind the ones numbered so far, then the exact sweep SAH from their leaves
    std::vector<int> extra_base(closed.size());
    int extra = next_id;
    for (size_t j = 0; j < closed.size(); j++) { extra_base[j] = extra; extra += closed[j].count - 2; }
    if (extra != n_int) return GLRT_HOST_EINVAL;  // (cannot happen: a binary tree over n leaves has n - 1 internal nodes)
    std::vector<std::vector<int>> members(closed.size());
    for (int k = 0; k < n; k++)
        if (closed_of[(size_t)k] >= 0) members[(size_t)closed_of[(size_t)k]].push_back(n_int + k);  // ascending
    std::vector<int> slots;
    for (size_t j = 0; j < closed.size(); j++) {
        slots.clear();
        for (int q = 0; q < closed[j].count - 2; q++) slots.push_back(extra_base[j] + q);
        lbvh::rebuild_one(nodes_out, closed[j].root, slots, members[j]);
    }
    // boxes of the nodes above the closed subtrees: children have larger numbers than their parents (breadth-first numbering)
    std::vector<char> is_closed_root((size_t)n_int, 0);
    for (auto &c : closed) is_closed_root[(size_t)c.root] = 1;
    for (int i = next_id - 1; i >= 0; i--) {
        if (is_closed_root[(size_t)i]) continue;
        lbvh::TBox b;
        b.reset();
        for (int k = 6; k <= 7; k++) {
            const float *C = nodes_out + 9 * (size_t)(int)nodes_out[9 * (size_t)i + k];
            b.grow(C, C + 3);
        }
        float *N = nodes_out + 9 * (size_t)i;
        for (int a = 0; a < 3; a++) { N[a] = b.lo[a]; N[3 + a] = b.hi[a]; }
    }
    const int max_depth = lbvh::tree_depth(nodes_out, n);
    if (max_depth_out) *max_depth_out = max_depth;
    return max_depth < 63 ? GLRT_HOST_OK : GLRT_HOST_EDEPTH;
}

// ---------------------------------------------------------------------------------------------- light side first
// The reference's traversal visits children.y first at every fork, whatever the ray (raytrace.frag:299-307: push x, push y, pop y).  Which child is which is the
// BUILDER's choice -- and half of all rays are shadow rays, every one of them aimed at a light: its search is over as soon as the light is hit (every box beyond is
// culled by tHit from then on), but until then nothing culls, and at a fork that holds the light in its x child the ray first walks the whole y subtree.  So, after any
// of the builders: at every fork where exactly ONE child's subtree contains emitting triangles, that child goes into the y slot.  (Emitting: the triangle's material has
// a non-zero emission -- the rule that puts a triangle on the light list, scene.cpp:246-248.)  Boxes, subtrees and node numbers stay; two child references are exchanged.
// On the headline scene two forks near the root are exchanged and the frame is 4 % shorter (profiles/r05_lights_first.txt); closest-hit results do not depend on the
// order (exact ties aside, as with any difference between two builders' trees: SURVEY.md H4).
// nodes: the wire format (9 floats per node, root = node 0), modified in place.  Returns the number of forks exchanged, or a negative GLRT_HOST_E* code.
int glrt_bvh_lights_first(float *nodes, size_t n_nodes, const float *tri, size_t n_tri, const float *mat, size_t n_mat) {
    if (!nodes || n_nodes == 0) return 0;
    if (!tri || !mat) return GLRT_HOST_EINVAL;
    std::vector<char> emits(n_mat, 0);
    for (size_t m = 0; m < n_mat; m++) {
        const float *e = mat + 18 * m + 3;  // texel 1 of the material record: emission
        emits[m] = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) != 0.0f;
    }
    // post-order over the tree (iterative; a child index out of range or met twice ends the pass: the upload reports such trees)
    std::vector<char> has_light(n_nodes, 0), seen(n_nodes, 0);
    std::vector<size_t> order, st{0};
    order.reserve(n_nodes);
    while (!st.empty()) {
        const size_t i = st.back();
        st.pop_back();
        if (i >= n_nodes || seen[i]) return GLRT_HOST_EINVAL;
        seen[i] = 1;
        order.push_back(i);
        const float *N = nodes + 9 * i;
        if (N[8] < 0.0f)
            for (int k = 6; k <= 7; k++)
                if (N[k] >= 0.0f) st.push_back((size_t)N[k]);  // (a child of -1 is absent: bvh.cpp:73-75)
    }
    int swapped = 0;
    for (size_t q = order.size(); q-- > 0;) {
        const size_t i = order[q];
        float *N = nodes + 9 * i;
        if (N[8] >= 0.0f) {
            const size_t t = (size_t)N[8];
            if (t < n_tri) {
                const float fm = tri[4 * t + 3];
                has_light[i] = fm >= 0.0f && (size_t)fm < n_mat && emits[(size_t)fm];
            }
            continue;
        }
        const bool lx = N[6] >= 0.0f && has_light[(size_t)N[6]], ly = N[7] >= 0.0f && has_light[(size_t)N[7]];
        has_light[i] = lx || ly;
        if (lx && !ly && N[7] >= 0.0f) { std::swap(N[6], N[7]); swapped++; }
    }
    return swapped;
}

// The order of a fork's children from MEASURED hits (round 6): tri_hits[t] = how often triangle t was a path ray's closest hit in a calibration frame
// (glrtx_hit_histogram: the render kernel counts them while it shades).  At every fork with two children the one whose subtree collected more hits goes into the y slot --
// the one the reference's traversal visits first (raytrace.frag:299-307): a ray that finds its hit there culls the sibling's box, or most of what is inside it.  Forks whose
// subtrees were hit equally often (no ray came there at all, usually) keep the builder's order.  Like glrt_bvh_lights_first this changes no box and no closest hit; which
// of two EXACTLY tied triangles a ray reports can change.  Apply it after every other pass (glrt_bvh_lights_first included: where one child holds the
// only lights it usually also collects the hits -- and if it does not, the measured order is the better one for the rays that were counted).
// Returns the number of forks whose children were exchanged, or a negative GLRT_HOST_E* code for a malformed tree.
int glrt_bvh_order_by_hits(float *nodes, size_t n_nodes, const uint32_t *tri_hits, size_t n_tri) {
    if (!nodes || n_nodes == 0) return 0;
    if (!tri_hits) return GLRT_HOST_EINVAL;
    std::vector<char> seen(n_nodes, 0);
    std::vector<size_t> order, st{0};
    order.reserve(n_nodes);
    while (!st.empty()) {
        const size_t i = st.back();
        st.pop_back();
        if (i >= n_nodes || seen[i]) return GLRT_HOST_EINVAL;
        seen[i] = 1;
        order.push_back(i);
        const float *N = nodes + 9 * i;
        if (N[8] < 0.0f)
            for (int k = 6; k <= 7; k++)
                if (N[k] >= 0.0f) st.push_back((size_t)N[k]);
    }
    std::vector<uint64_t> hits(n_nodes, 0);
    std::vector<uint32_t> leaves(n_nodes, 0);
    // Hits PER UNIT COST decide (a subtree of n triangles is charged n^e, e = 0.5): which of two searches to make first, when the first one's success spares the second,
    // is a question of probability over cost, not of probability alone.  Hits alone (e = 0, this pass's first form) sent every ray through the bigger child first and cost
    // config 5 +11.6 % on a less balanced tree; with e = 0.5 / 1: config 5 -1.0 %, config 2 -1.0 / -0.7 %, headline -0.2 / -0.3 %, config 4 +1.4 % (hits alone: -0.75 %),
    // profiles/r06_hit_order_cost.txt.  GLRT_HITS_COST_EXP overrides (A/B).
    const char *ee = std::getenv("GLRT_HITS_COST_EXP");
    const double e = ee ? std::max(0.0, std::atof(ee)) : 0.5;
    int swapped = 0;
    for (size_t q = order.size(); q-- > 0;) {  // children before parents
        const size_t i = order[q];
        float *N = nodes + 9 * i;
        if (N[8] >= 0.0f) {
            const size_t t = (size_t)N[8];
            hits[i] = t < n_tri ? tri_hits[t] : 0;
            leaves[i] = 1;
            continue;
        }
        const bool hx = N[6] >= 0.0f, hy = N[7] >= 0.0f;
        const uint64_t ax = hx ? hits[(size_t)N[6]] : 0, ay = hy ? hits[(size_t)N[7]] : 0;
        const uint32_t cx = hx ? leaves[(size_t)N[6]] : 0, cy = hy ? leaves[(size_t)N[7]] : 0;
        hits[i] = ax + ay;
        leaves[i] = cx + cy;
        const bool x_first = e == 0.0 ? ax > ay : (double)ax * std::pow((double)std::max(cy, 1u), e) > (double)ay * std::pow((double)std::max(cx, 1u), e);
        if (hx && hy && x_first) { std::swap(N[6], N[7]); swapped++; }
    }
    return swapped;
}

// glrtx_hit_histogram counts the closest hits of PATH rays.  Every hit that is shaded also sends a shadow ray to a light triangle drawn uniformly (raytrace.frag:341-343),
// and half of all rays are such rays: their share is added here by that rule -- (all counted hits) / (number of light triangles) to every emitting triangle -- so that
// glrt_bvh_order_by_hits weighs the side of a fork the shadow rays end in as well (on a Cornell-box scene they are what decides: without them the measured order undid
// glrt_bvh_lights_first and the frame took 2-4 % longer).  Returns the number of emitting triangles.
int glrt_bvh_add_shadow_hits(uint32_t *tri_hits, size_t n_tri, const float *tri, const float *mat, size_t n_mat) {
    if (!tri_hits || !tri || !mat) return GLRT_HOST_EINVAL;
    std::vector<char> emits(n_mat, 0);
    for (size_t m = 0; m < n_mat; m++) {
        const float *e = mat + 18 * m + 3;
        emits[m] = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) != 0.0f;
    }
    uint64_t total = 0;
    size_t n_light = 0;
    for (size_t t = 0; t < n_tri; t++) {
        total += tri_hits[t];
        const float fm = tri[4 * t + 3];
        if (fm >= 0.0f && (size_t)fm < n_mat && emits[(size_t)fm]) n_light++;
    }
    if (n_light == 0) return 0;
    const uint64_t share = total / n_light;
    for (size_t t = 0; t < n_tri; t++) {
        const float fm = tri[4 * t + 3];
        if (fm >= 0.0f && (size_t)fm < n_mat && emits[(size_t)fm]) tri_hits[t] = (uint32_t)std::min<uint64_t>(UINT32_MAX, (uint64_t)tri_hits[t] + share);
    }
    return (int)n_light;
}

// ---- Reinsertion, an optimisation pass over a finished tree (after Bittner, Hapala, Havran, "Fast insertion-based optimization of bounding volume hierarchies", CGF 2013,
// in the per-node form of Meister & Bittner 2018).  A top-down SAH build decides every split once, greedily; afterwards each subtree N is taken out (its parent's slot P is
// freed, the sibling moves up) and put back where the tree's summed box area grows least: a branch-and-bound search over all positions X, cost(X) = area(X u N) + the
// growth of X's ancestors, candidates ordered by that growth, a branch abandoned once growth + area(N) cannot beat the best position found.  The old position is among
// the candidates, so the sum of the forks' areas -- the expected number of box visits of a ray that culls nothing, which is what the reference's fixed-order traversal
// does until its first hit -- never rises.  Passes repeat until one gains less than 0.1 %.  The children of every fork are then put in a top-down builder's order and the
// result is renumbered in DFS pre-order (root = node 0, x child first).  Measured: config 5 -2.9 % per frame (its fork area: -2.9 %), config 4 -0.6 %, headline -0.1 %.
// Closest-hit results do not depend on the tree (exact ties and grazing-ray box misses aside: SURVEY.md H4).
// nodes: wire format, modified in place.  cost_out (may be NULL): [0] the summed fork area / root area before, [1] after.  Returns the number of subtrees that moved,
// 0 for trees it leaves alone (fewer than 4 leaves, absent children, non-finite boxes), or a negative GLRT_HOST_E* code for a malformed tree.
int glrt_bvh_reinsert(float *nodes, size_t n_nodes, int max_passes, int *max_depth_out, double *cost_out) {
    if (!nodes || n_nodes == 0) return GLRT_HOST_EINVAL;
    const int n = (int)n_nodes;
    const std::vector<float> original(nodes, nodes + 9 * n_nodes);  // (the passes work in place)
    std::vector<int> parent((size_t)n, -1);
    auto L = [&](int i) -> float * { return nodes + 9 * (size_t)i; };
    auto is_fork = [&](int i) { return L(i)[8] < 0.0f; };
    {
        std::vector<char> seen((size_t)n, 0);
        std::vector<int> st{0};
        size_t met = 0;
        bool leave = n < 7;
        while (!st.empty()) {
            const int i = st.back();
            st.pop_back();
            if (i < 0 || i >= n || seen[(size_t)i]) return GLRT_HOST_EINVAL;
            seen[(size_t)i] = 1;
            met++;
            for (int k = 0; k < 6; k++)
                if (!(L(i)[k] - L(i)[k] == 0.0f)) leave = true;
            if (is_fork(i))
                for (int k = 6; k <= 7; k++) {
                    const float c = L(i)[k];
                    if (c < 0.0f) { leave = true; continue; }
                    if (!(c < (float)n)) return GLRT_HOST_EINVAL;
                    parent[(size_t)c] = i;
                    st.push_back((int)c);
                }
        }
        if (met != (size_t)n) leave = true;  // (unreachable records: not ours to renumber)
        if (leave) {
            if (cost_out) cost_out[0] = cost_out[1] = 0.0;
            if (max_depth_out) *max_depth_out = -1;
            return 0;
        }
    }
    auto area = [&](int i) -> double {
        const float *b = L(i);
        const double dx = (double)b[3] - b[0], dy = (double)b[4] - b[1], dz = (double)b[5] - b[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto union_area = [&](int x, const float *nb) -> double {
        const float *b = L(x);
        const double dx = (double)std::max(b[3], nb[3]) - std::min(b[0], nb[0]), dy = (double)std::max(b[4], nb[4]) - std::min(b[1], nb[1]),
                     dz = (double)std::max(b[5], nb[5]) - std::min(b[2], nb[2]);
        return dx * dy + dy * dz + dz * dx;
    };
    auto refit_up = [&](int i) {  // boxes of i and its ancestors from their children, until one does not change
        for (; i >= 0; i = parent[(size_t)i]) {
            float *b = L(i);
            const float *p = L((int)b[6]), *q = L((int)b[7]);
            bool changed = false;
            for (int k = 0; k < 3; k++) {
                const float lo = std::min(p[k], q[k]), hi = std::max(p[3 + k], q[3 + k]);
                if (lo != b[k] || hi != b[3 + k]) changed = true;
                b[k] = lo; b[3 + k] = hi;
            }
            if (!changed) break;
        }
    };
    auto total_cost = [&]() {
        double s = 0.0;
        for (int i = 0; i < n; i++)
            if (is_fork(i)) s += area(i);
        return s / std::max(area(0), 1e-300);
    };
    const double cost0 = total_cost();
    double cost_prev = cost0;
    int moved_total = 0;
    std::vector<int> order;
    typedef std::pair<double, int> QE;
    std::vector<QE> heap;
    for (int pass = 0; pass < max_passes; pass++) {
        order.clear();
        for (int i = 1; i < n; i++) order.push_back(i);
        std::vector<double> key((size_t)n);
        for (int i = 0; i < n; i++) key[(size_t)i] = area(i);
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return key[(size_t)x] > key[(size_t)y]; });
        int moved = 0;
        for (const int N : order) {
            const int P = parent[(size_t)N];
            if (P <= 0) continue;  // (the root's children stay: the root keeps slot 0)
            const int G = parent[(size_t)P];
            const int S = (int)L(P)[6] == N ? (int)L(P)[7] : (int)L(P)[6];
            // take N out: S takes P's place under G
            L(G)[(int)L(G)[6] == P ? 6 : 7] = (float)S;
            parent[(size_t)S] = G;
            refit_up(G);
            // the best place: X becomes the sibling of N under P
            float nb[6];
            for (int k = 0; k < 6; k++) nb[k] = L(N)[k];
            const double an = area(N);
            double best = std::numeric_limits<double>::infinity();
            int best_x = S;
            heap.clear();
            const double root_growth = union_area(0, nb) - area(0);
            heap.push_back(QE(-root_growth, (int)L(0)[6]));
            heap.push_back(QE(-root_growth, (int)L(0)[7]));
            std::make_heap(heap.begin(), heap.end());
            while (!heap.empty()) {
                std::pop_heap(heap.begin(), heap.end());
                const QE e = heap.back();
                heap.pop_back();
                const double induced = -e.first;
                if (induced + an >= best) break;
                const int X = e.second;
                const double direct = union_area(X, nb);
                if (induced + direct < best) { best = induced + direct; best_x = X; }
                if (is_fork(X)) {
                    const double below = induced + (direct - area(X));
                    if (below + an < best) {
                        heap.push_back(QE(-below, (int)L(X)[6])); std::push_heap(heap.begin(), heap.end());
                        heap.push_back(QE(-below, (int)L(X)[7])); std::push_heap(heap.begin(), heap.end());
                    }
                }
            }
            const int X = best_x, XP = parent[(size_t)X];
            L(XP)[(int)L(XP)[6] == X ? 6 : 7] = (float)P;
            parent[(size_t)P] = XP;
            L(P)[6] = (float)X; L(P)[7] = (float)N; L(P)[8] = -1.0f;
            parent[(size_t)X] = P;
            parent[(size_t)N] = P;
            for (int k = 0; k < 3; k++) { L(P)[k] = std::min(L(X)[k], nb[k]); L(P)[3 + k] = std::max(L(X)[3 + k], nb[3 + k]); }
            refit_up(XP);
            if (X != S) moved++;
        }
        moved_total += moved;
        const double c = total_cost();
        const bool done = moved == 0 || c > cost_prev * 0.999;
        cost_prev = c;
        if (done) break;
    }
    // the children of every fork in the order a top-down builder leaves them (a reinserted pair is in no particular order, and the order is worth as much as the
    // areas: profiles/r05_reinsert.txt): on the axis where the two boxes' centres differ most, the lower one is x
    for (int i = 0; i < n; i++) {
        if (!is_fork(i)) continue;
        const float *X = L((int)L(i)[6]), *Y = L((int)L(i)[7]);
        int k = 0;
        float dmax = -1.0f, dk = 0.0f;
        for (int a = 0; a < 3; a++) {
            const float d = 0.5f * (X[a] + X[3 + a]) - 0.5f * (Y[a] + Y[3 + a]);
            if (std::fabs(d) > dmax) { dmax = std::fabs(d); dk = d; k = a; }
        }
        (void)k;
        if (dk > 0.0f) std::swap(L(i)[6], L(i)[7]);
    }
    // renumber: DFS pre-order, x child first
    std::vector<float> out((size_t)n * 9);
    std::vector<int> new_of((size_t)n, -1);
    int next = 0, max_depth = 0;
    std::vector<std::pair<int, int>> st{{0, 0}};
    std::vector<int> old_of((size_t)n);
    while (!st.empty()) {
        const auto [i, d] = st.back();
        st.pop_back();
        new_of[(size_t)i] = next;
        old_of[(size_t)next++] = i;
        max_depth = std::max(max_depth, d);
        if (is_fork(i)) { st.push_back({(int)L(i)[7], d + 1}); st.push_back({(int)L(i)[6], d + 1}); }
    }
    for (int j = 0; j < n; j++) {
        const float *src = L(old_of[(size_t)j]);
        float *dst = out.data() + 9 * (size_t)j;
        std::memcpy(dst, src, 9 * sizeof(float));
        if (src[8] < 0.0f) { dst[6] = (float)new_of[(size_t)src[6]]; dst[7] = (float)new_of[(size_t)src[7]]; }
    }
    if (cost_out) { cost_out[0] = cost0; cost_out[1] = cost_prev; }
    if (max_depth_out) *max_depth_out = max_depth;
    // Insertion can deepen the tree.  Like every builder here the pass refuses a result the 64-entry traversal stack cannot hold (it used to hand it on, and
    // the upload rejected it seconds later without naming the pass): the caller's tree is put back as it came in.
    if (max_depth >= 63) {
        std::memcpy(nodes, original.data(), original.size() * sizeof(float));
        return GLRT_HOST_EDEPTH;
    }
    std::memcpy(nodes, out.data(), out.size() * sizeof(float));
    return moved_total;
}

// Chain ("brute force") tree: fork i has the global bounds and children
// (next fork, leaf i); the last fork holds the last two leaves.  The reference
// traversal order (push x, push y, pop y first; raytrace.frag:299-307) then
// visits leaf 0, 1, 2, ... with a stack depth of 2: a linear scan over all
// triangles, none culled.  Layout: node 2i = fork i, node 2i+1 = leaf i
// (DFS pre-order is not meaningful here); 2n-1 nodes like any full tree.
int glrt_bvh_build_chain(const float *vert, size_t n_vert, const float *tri, size_t n_tri, float *nodes_out) {
    if (!vert || !tri || !nodes_out || n_tri == 0) return GLRT_HOST_EINVAL;
    std::vector<Prim> prims;
    if (!load_prims(vert, n_vert, tri, n_tri, prims)) return GLRT_HOST_EINDEX;
    Box all;
    all.reset();
    for (auto &p : prims) all.grow(p.box);
    auto put = [&](size_t idx, const Box &b, float cx, float cy, float cz) {
        float *n = nodes_out + 9 * idx;
        n[0] = b.lo[0]; n[1] = b.lo[1]; n[2] = b.lo[2];
        n[3] = b.hi[0]; n[4] = b.hi[1]; n[5] = b.hi[2];
        n[6] = cx; n[7] = cy; n[8] = cz;
    };
    if (n_tri == 1) {
        put(0, prims[0].box, -1.f, -1.f, 0.f);
        return GLRT_HOST_OK;
    }
    // forks 0..n-2 at nodes 2i, leaves 0..n-2 at nodes 2i+1, last leaf at node 2n-2
    for (size_t i = 0; i + 1 < n_tri; i++) {
        const bool last = (i + 2 == n_tri);
        const float next = last ? (float)(2 * n_tri - 2) : (float)(2 * (i + 1));
        // children.x is pushed first and popped last: it must be the continuation
        put(2 * i, all, next, (float)(2 * i + 1), -1.f);
        put(2 * i + 1, prims[i].box, -1.f, -1.f, (float)i);
    }
    put(2 * n_tri - 2, prims[n_tri - 1].box, -1.f, -1.f, (float)(n_tri - 1));
    return GLRT_HOST_OK;
}

// ---------------------------------------------------------------------------------------------- the reference host's own tree
// glrt_bvh_build_reference: the tree the reference's host would hand its shader for these triangles -- BVH::construct / constructRec (src/core/bvh.cpp:59-160) with the
// Bounds / TriangleInfo arithmetic of bvh.h:11-82 -- restated, for hosts that want the reference's OWN choices where the image depends on the tree: which of two exactly
// tied triangles a ray reports, and at which pixels a grazing ray misses a flat box.  The other builders here make better trees; this one makes that one.
// The rule (every rounding as the reference's expressions perform it):
//   * a triangle's box is min / max over its three vertices, its centroid ((v0 + v1) + v2) / 3.0f per component (bvh.h:58-75);
//   * a node's box is the union of its triangles' boxes, accumulated from +-1e8, not +-inf (bvh.h:12-15): coordinates beyond 1e8 are clipped exactly as there;
//   * the split axis is the widest axis of the centroids' box, x before y before z at equal spans (bvh.h:36-44);
//   * up to 8 triangles: std::nth_element at (left + right) / 2 by centroid[axis] (:100-104);
//   * more: 16 buckets along that axis ONLY -- the bucket of a centroid from the FLOAT difference to the box's minimum, widened to double, times
//     16 / (|max - min| + 1e-8) in double (:110-121); cost of splitting behind bucket i = 0.125 + float((n0 area0 + n1 area1) / area) (:123-136), the first minimum
//     wins (:138-145); if that cost is below the triangle count the range is std::partition-ed by a predicate that computes the bucket AGAIN, this time from the
//     DOUBLE difference (bvh.cpp:39-47: the two can disagree in the last place at a bucket's edge -- restated as written); otherwise the range is cut at its middle
//     in the order it has, unsorted (:147-152 leaves `mid` alone);
//   * nodes in pre-order, the left subtree first (:77-78, :155-157); a leaf holds one triangle.
// nth_element and partition are the standard library's own (as in the reference): the order they leave equal and unordered elements in is this libstdc++'s, which is
// the reference's when it is built with the same toolchain.  PARITY UNPINNED: the reference's host cannot be built here, so no tree of its making exists
// to compare with; tests/test_host.py checks the rule level by level against a numpy statement of the same arithmetic.
// Where the reference itself has no defined outcome this builder stays defined: a bucket index that is not a number in range goes to bucket 0 / 15 (the reference
// indexes out of bounds), a NaN centroid is ordered as 0 by the nth_element, and a partition that leaves one side empty (the reference recurses for ever) is replaced by the cut at the middle.
namespace reftree {
struct Extent {  // bvh.h:11-56
    float lo[3] = {1.0e8f, 1.0e8f, 1.0e8f}, hi[3] = {-1.0e8f, -1.0e8f, -1.0e8f};
    void take(const float *p) {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
    }
    void take(const Extent &o) {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], o.lo[a]); hi[a] = std::max(hi[a], o.hi[a]); }
    }
    int widest() const {
        const float sx = std::abs(hi[0] - lo[0]), sy = std::abs(hi[1] - lo[1]), sz = std::abs(hi[2] - lo[2]);
        const float m = std::max(sx, std::max(sy, sz));
        return m == sx ? 0 : (m == sy ? 1 : 2);
    }
    float area() const {
        const float sx = std::abs(hi[0] - lo[0]), sy = std::abs(hi[1] - lo[1]), sz = std::abs(hi[2] - lo[2]);
        return 2.0f * (sx * sy + sy * sz + sz * sx);
    }
};
struct Face {
    int index;
    float c[3];
    Extent box;
};
constexpr int kBuckets = 16;
inline int clamp_bucket(double x) {  // static_cast<int> of the reference, kept in range where that cast is undefined
    if (!(x >= 0.0)) return 0;
    return x >= (double)kBuckets ? kBuckets : (int)x;
}
}  // namespace reftree

int glrt_bvh_build_reference(const float *vert, size_t n_vert, const float *tri, size_t n_tri, float *nodes_out, int *max_depth_out) {
    using namespace reftree;
    if (!vert || !tri || !nodes_out || n_tri == 0) return GLRT_HOST_EINVAL;
    if (n_tri > (size_t)1 << 29) return GLRT_HOST_EINVAL;
    std::vector<Face> f(n_tri);
    for (size_t t = 0; t < n_tri; t++) {
        const float *v[3];
        for (int k = 0; k < 3; k++) {
            const float fi = tri[4 * t + k];
            if (!(fi >= 0.f) || (size_t)fi >= n_vert) return GLRT_HOST_EINDEX;
            v[k] = vert + GLRT_VERTEX_FLOATS * (size_t)fi;
        }
        Face &x = f[t];
        x.index = (int)t;
        for (int a = 0; a < 3; a++) {
            x.box.lo[a] = std::min(v[0][a], std::min(v[1][a], v[2][a]));
            x.box.hi[a] = std::max(v[0][a], std::max(v[1][a], v[2][a]));
            x.c[a] = (v[0][a] + v[1][a] + v[2][a]) / 3.0f;
        }
    }
    struct Job { int l, r, node, depth; };
    std::vector<Job> todo;
    todo.push_back({0, (int)n_tri, 0, 0});
    int deepest = 0;
    while (!todo.empty()) {
        const Job j = todo.back();
        todo.pop_back();
        const int l = j.l, r = j.r, count = r - l;
        Extent all;
        for (int i = l; i < r; i++) all.take(f[i].box);
        float *out = nodes_out + 9 * (size_t)j.node;
        for (int a = 0; a < 3; a++) { out[a] = all.lo[a]; out[3 + a] = all.hi[a]; }
        deepest = std::max(deepest, j.depth);
        if (count == 1) {
            out[6] = -1.f; out[7] = -1.f; out[8] = (float)f[l].index;
            continue;
        }
        Extent cb;
        for (int i = l; i < r; i++) cb.take(f[i].c);
        const int axis = cb.widest();
        int mid = (l + r) / 2;
        if (count <= 8) {
            // (a centroid that is not a number is ordered as 0: with the bare `<` of :12-14 such input breaks the strict weak order nth_element relies on -- undefined in the reference)
            auto key = [axis](const Face &x) { const float c = x.c[axis]; return c == c ? c : 0.0f; };
            std::nth_element(f.begin() + l, f.begin() + mid, f.begin() + r, [&key](const Face &x, const Face &y) { return key(x) < key(y); });
        } else {
            int cnt[kBuckets] = {0};
            Extent bb[kBuckets];
            const double cmin = cb.lo[axis], cmax = cb.hi[axis];
            const double per_span = 1.0 / (std::abs(cmax - cmin) + 1.0e-8);  // (:112 and :42: the same value in both places)
            for (int i = l; i < r; i++) {
                const float numer_f = f[i].c[axis] - cb.lo[axis];  // a float difference, widened afterwards (:113)
                const double numer = numer_f;
                int b = clamp_bucket(kBuckets * std::abs(numer) * per_span);
                if (b == kBuckets) b = kBuckets - 1;
                cnt[b]++;
                bb[b].take(f[i].box);
            }
            double cost[kBuckets - 1];
            for (int i = 0; i < kBuckets - 1; i++) {
                Extent b0, b1;
                int c0 = 0, c1 = 0;
                for (int k = 0; k <= i; k++) { b0.take(bb[k]); c0 += cnt[k]; }
                for (int k = i + 1; k < kBuckets; k++) { b1.take(bb[k]); c1 += cnt[k]; }
                const float ratio = ((float)c0 * b0.area() + (float)c1 * b1.area()) / all.area();
                cost[i] = 0.125 + (double)ratio;
            }
            double best = cost[0];
            int split = 0;
            for (int i = 1; i < kBuckets - 1; i++)
                if (best > cost[i]) { best = cost[i]; split = i; }
            if (best < (double)count) {
                auto it = std::partition(f.begin() + l, f.begin() + r, [&](const Face &x) {
                    const double diff = std::abs((double)x.c[axis] - cmin);  // a double difference this time (:43)
                    int b = clamp_bucket(kBuckets * diff * per_span);
                    if (b >= kBuckets) b = kBuckets - 1;
                    return b <= split;
                });
                const int m = (int)(it - f.begin());
                if (m != l && m != r) mid = m;  // (an empty side: the reference would never return)
            }
        }
        const int left = j.node + 1, right = j.node + 2 * (mid - l);  // a subtree over k triangles has 2k - 1 nodes
        out[6] = (float)left; out[7] = (float)right; out[8] = -1.f;
        todo.push_back({mid, r, right, j.depth + 1});
        todo.push_back({l, mid, left, j.depth + 1});
    }
    if (max_depth_out) *max_depth_out = deepest;
    return deepest < 63 ? GLRT_HOST_OK : GLRT_HOST_EDEPTH;
}

}  // extern "C"
