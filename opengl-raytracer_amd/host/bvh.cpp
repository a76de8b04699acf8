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
// Closest-hit results do not depend on the tree shape (only exact ties do), so
// any valid tree is parity-equivalent.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

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

constexpr int kBins = 16;
// Past this depth the builder switches to median splits, which bounds the total
// depth by kSahDepthCap + ceil(log2(n)) (< 64 for n < 2^23: the reference
// shader's stack is int[64], raytrace.frag:284).
constexpr int kSahDepthCap = 40;

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
                    int k = std::min(kBins - 1, (int)((prims[i].c[a] - cb.lo[a]) * scale));
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
                    float cost = acc.half_area() * (float)c + right_area[k + 1] * (float)right_cnt[k + 1];
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
                    return std::min(kBins - 1, (int)((p.c[a] - lo) * scale)) <= best_bin;
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
        for (int a = 0; a < 3; a++) p.c[a] = 0.5f * (p.box.lo[a] + p.box.hi[a]);
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

}  // extern "C"
