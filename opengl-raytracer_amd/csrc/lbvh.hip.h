// lbvh.hip.h -- linear BVH construction on the device (BASELINE config 5).
//
// Replaces, for large scenes, the role of the reference's CPU builder BVH::construct
// (src/core/bvh.cpp:59-160) and writes the same flat 'u_bvhBuffer' node format (bvh.h:84-100):
// node = {bboxMin, bboxMax, children}, fork children = (x, y, -1), leaf = (-1, -1, triangle).
// Algorithm: Karras 2012 -- 30-bit Morton code of each triangle box centre, made unique by appending the
// triangle index, radix-sorted (hipCUB); every internal node finds its key range and split from the
// common-prefix lengths of neighbouring keys, independently; boxes are fitted bottom-up, one launch per tree level; five
// sweeps of tree rotations (one launch per level and sweep) then improve the Morton tree's surface-area cost.
// Layout: internal node i at index i (root = 0), the leaf of sorted
// position k at index (n - 1) + k.  Every step is integer arithmetic or a single correctly rounded float
// operation, so the result equals glrt_bvh_build_lbvh (host/bvh.cpp, the CPU statement) bit for bit.
// Tree shape never changes what the path tracer computes (only exact ties).
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>

namespace glrtx {
namespace lbvh {

__device__ __forceinline__ unsigned expand10(unsigned v) {  // 10 bits -> every third bit
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__device__ __forceinline__ unsigned quantize(float c, float lo, float ext) {
    if (!(ext > 0.0f)) return 0u;
    const float q = (c - lo) / ext * 1024.0f;
    if (!(q >= 0.0f)) return 0u;  // (also a NaN: an extent that overflowed)
    return q >= 1024.0f ? 1023u : (unsigned)(int)q;
}
// A box's centre as the build orders by it: 0 where it is not finite (host/bvh.cpp: centre)
__device__ __forceinline__ float centre(float lo, float hi) {
    const float c = 0.5f * (lo + hi);
    return (c - c == 0.0f) ? c : 0.0f;
}

// float <-> unsigned with the same ordering, for atomicMin/atomicMax
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

struct Box3 {
    float lo[3], hi[3];
};
// std::min / std::max as the CPU builder evaluates them (matters only for the sign of a zero)
__device__ __forceinline__ float min_std(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float max_std(float a, float b) { return (a < b) ? b : a; }
// load that sees what another CU wrote before its device-scope fence
__device__ __forceinline__ float load_coherent(const float *p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Triangle box from the wire-format buffers; returns false if a vertex index is out of range.
__device__ __forceinline__ bool tri_box(const float *vert, unsigned n_vert, const float *tri, unsigned t, Box3 &b) {
    for (int a = 0; a < 3; a++) { b.lo[a] = __builtin_inff(); b.hi[a] = -__builtin_inff(); }
    for (int k = 0; k < 3; k++) {
        const float fi = tri[4 * (size_t)t + k];
        if (!(fi >= 0.0f) || (unsigned)fi >= n_vert) return false;
        const float *p = vert + 15 * (size_t)(unsigned)fi;
        for (int a = 0; a < 3; a++) {
            b.lo[a] = min_std(b.lo[a], p[a]);
            b.hi[a] = max_std(b.hi[a], p[a]);
        }
    }
    return true;
}

// bounds[0..2] = min, [3..5] = max of the box centres (ordered-unsigned encoding); err = bad index flag
__global__ __launch_bounds__(256) void k_centre_bounds(const float *vert, unsigned n_vert, const float *tri, unsigned n, unsigned *bounds,
                                                       int *err) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    unsigned lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    if (t < n) {
        Box3 b;
        if (!tri_box(vert, n_vert, tri, t, b)) *err = 1;
        else
            for (int a = 0; a < 3; a++) lo[a] = hi[a] = f2ord(centre(b.lo[a], b.hi[a]));
    }
    // one atomic per workgroup and component, each component on its own 128-byte line (atomics on one line serialise
    // at ~90 per microsecond)
    __shared__ unsigned red[4][6];
    for (int a = 0; a < 3; a++) {
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned l2 = (unsigned)__shfl_xor((int)lo[a], m), h2 = (unsigned)__shfl_xor((int)hi[a], m);
            lo[a] = l2 < lo[a] ? l2 : lo[a];
            hi[a] = h2 > hi[a] ? h2 : hi[a];
        }
        if ((threadIdx.x & 63u) == 0u) { red[threadIdx.x >> 6][a] = lo[a]; red[threadIdx.x >> 6][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6u) {
        const unsigned a = threadIdx.x;
        unsigned v = red[0][a];
        for (int w = 1; w < 4; w++) v = a < 3u ? (red[w][a] < v ? red[w][a] : v) : (red[w][a] > v ? red[w][a] : v);
        if (a < 3u) atomicMin(&bounds[32 * a], v);
        else atomicMax(&bounds[32 * a], v);
    }
}

__global__ __launch_bounds__(256) void k_keys(const float *vert, unsigned n_vert, const float *tri, unsigned n, const unsigned *bounds,
                                              unsigned long long *keys) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    Box3 b;
    if (!tri_box(vert, n_vert, tri, t, b)) { keys[t] = (unsigned long long)t; return; }
    unsigned m = 0;
    for (int a = 0; a < 3; a++) {
        const float lo = ord2f(bounds[32 * a]), hi = ord2f(bounds[32 * (3 + a)]);
        const float c = centre(b.lo[a], b.hi[a]);
        m |= expand10(quantize(c, lo, hi - lo)) << (2 - a);
    }
    keys[t] = ((unsigned long long)m << 32) | t;
}

__device__ __forceinline__ int delta(const unsigned long long *k, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(k[i] ^ k[j]));  // keys are unique: the xor is never 0
}

// One thread per internal node: children (>= 0 internal, < 0 ~leaf position) into nodes[i].children as node
// indices, and the parent links.
__global__ __launch_bounds__(256) void k_hierarchy(const unsigned long long *keys, int n, float *nodes, int *parent) {
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int g = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int left = (lo == g) ? (n - 1) + g : g;          // node indices
    const int right = (hi == g + 1) ? (n - 1) + g + 1 : g + 1;
    float *o = nodes + 9 * (size_t)i;
    o[6] = (float)left; o[7] = (float)right; o[8] = -1.0f;
    parent[left] = i;
    parent[right] = i;
    if (i == 0) parent[0] = -1;
}

// One thread per leaf: write the leaf node; the tree depth (deepest leaf) comes out as a by-product, one atomic per wave.
__global__ __launch_bounds__(256) void k_leaves(const float *vert, unsigned n_vert, const float *tri, const unsigned long long *keys, int n,
                                                float *nodes, const int *parent, int *max_depth) {
    const int k = (int)(blockIdx.x * 256u + threadIdx.x);
    int depth = 0;
    if (k < n)
        for (int c = parent[(n - 1) + k]; c >= 0; c = parent[c]) depth++;
    for (int m = 32; m >= 1; m >>= 1) {
        const int o = __shfl_xor(depth, m);
        depth = o > depth ? o : depth;
    }
    if ((threadIdx.x & 63u) == 0u) atomicMax(max_depth, depth);
    if (k >= n) return;
    const unsigned t = (unsigned)keys[k];
    Box3 b;
    tri_box(vert, n_vert, tri, t, b);
    float *o = nodes + 9 * (size_t)((n - 1) + k);
    o[0] = b.lo[0]; o[1] = b.lo[1]; o[2] = b.lo[2];
    o[3] = b.hi[0]; o[4] = b.hi[1]; o[5] = b.hi[2];
    o[6] = -1.0f; o[7] = -1.0f; o[8] = (float)t;
}

// Bottom-up fit, one launch per tree level: in round r an internal node whose children were both finished in EARLIER
// rounds (leaves: from the start) takes the union of their boxes and stamps itself with r.  Reading only what earlier
// launches wrote needs no fences or atomics between workgroups, and a node of height h is fitted in round h, so
// `depth` rounds fit the whole tree.
__global__ __launch_bounds__(256) void k_fit_round(int n, int round, float *nodes, int *stamp) {
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n - 1 || stamp[i] != 0) return;
    float *p = nodes + 9 * (size_t)i;
    const int l = (int)p[6], r = (int)p[7];
    const int sl = l >= n - 1 ? -1 : stamp[l], sr = r >= n - 1 ? -1 : stamp[r];  // -1: a leaf
    if (sl == 0 || sr == 0 || sl >= round || sr >= round) return;
    const float *a = nodes + 9 * (size_t)l, *b = nodes + 9 * (size_t)r;
    for (int k = 0; k < 3; k++) {
        p[k] = min_std(a[k], b[k]);
        p[3 + k] = max_std(a[3 + k], b[3 + k]);
    }
    stamp[i] = round;
}

// ---- tree rotations (the quality pass; CPU statement and rationale: host/bvh.cpp, lbvh::rotate_tree) ----
// Original depth of every internal node (root 0), by walking up the parent links.
__global__ __launch_bounds__(256) void k_levels(int n, const int *parent, int *level) {
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n - 1) return;
    int d = 0;
    for (int c = parent[i]; c >= 0; c = parent[c]) d++;
    level[i] = d;
}
__device__ __forceinline__ float half_area9(const float *lo, const float *hi) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return (dx * dy + dy * dz) + dz * dx;
}
__device__ __forceinline__ float union_area9(const float *a, const float *b) {
    float lo[3], hi[3];
    for (int k = 0; k < 3; k++) { lo[k] = min_std(a[k], b[k]); hi[k] = max_std(a[3 + k], b[3 + k]); }
    return half_area9(lo, hi);
}
// One sweep step: every internal node that was at depth `d` when the sweep started tries its four rotations.  Such nodes have
// disjoint subtrees, a rotation rearranges only the subtree of its own node, and nothing above level d has moved yet in this
// sweep, so the threads of one launch never touch the same record.
__global__ __launch_bounds__(256) void k_rotate_level(int n, int d, const int *level, float *nodes, int *parent) {
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n - 1 || level[i] != d) return;
    float *N = nodes + 9 * (size_t)i;
    const int c[2] = {(int)N[6], (int)N[7]};
    float best = 0.0f;
    int kind = -1, bside = 0, bwhich = 0;  // kind 0: child <-> grandchild, 1: grandchild <-> grandchild
    for (int side = 0; side < 2; side++) {
        const int x = c[side], y = c[side ^ 1];
        if (x >= n - 1) continue;
        const float *X = nodes + 9 * (size_t)x, *Y = nodes + 9 * (size_t)y;
        const float old = half_area9(X, X + 3);
        const float g0 = union_area9(Y, nodes + 9 * (size_t)(int)X[7]) - old;
        const float g1 = union_area9(nodes + 9 * (size_t)(int)X[6], Y) - old;
        if (g0 < best) { best = g0; kind = 0; bside = side; bwhich = 0; }
        if (g1 < best) { best = g1; kind = 0; bside = side; bwhich = 1; }
    }
    if (c[0] < n - 1 && c[1] < n - 1) {
        const float *A = nodes + 9 * (size_t)c[0], *B = nodes + 9 * (size_t)c[1];
        const float *A1 = nodes + 9 * (size_t)(int)A[6], *A2 = nodes + 9 * (size_t)(int)A[7];
        const float *B1 = nodes + 9 * (size_t)(int)B[6], *B2 = nodes + 9 * (size_t)(int)B[7];
        const float old = half_area9(A, A + 3) + half_area9(B, B + 3);
        const float h0 = (union_area9(B1, A2) + union_area9(A1, B2)) - old;
        const float h1 = (union_area9(B2, A2) + union_area9(B1, A1)) - old;
        if (h0 < best) { best = h0; kind = 1; bwhich = 0; }
        if (h1 < best) { best = h1; kind = 1; bwhich = 1; }
    }
    if (kind < 0) return;
    auto refit = [&](float *X) {
        const float *P = nodes + 9 * (size_t)(int)X[6], *Q = nodes + 9 * (size_t)(int)X[7];
        for (int k = 0; k < 3; k++) { X[k] = min_std(P[k], Q[k]); X[3 + k] = max_std(P[3 + k], Q[3 + k]); }
    };
    if (kind == 0) {
        float *X = nodes + 9 * (size_t)c[bside];
        const int y = c[bside ^ 1];
        const int moved = (int)X[6 + bwhich];
        X[6 + bwhich] = (float)y;
        N[6 + (bside ^ 1)] = (float)moved;
        parent[y] = c[bside];
        parent[moved] = i;
        refit(X);
    } else {
        float *A = nodes + 9 * (size_t)c[0], *B = nodes + 9 * (size_t)c[1];
        const float a1 = A[6];
        A[6] = B[6 + bwhich];
        B[6 + bwhich] = a1;
        parent[(int)A[6]] = c[0];
        parent[(int)B[6 + bwhich]] = c[1];
        refit(A);
        refit(B);
    }
}
// Depth of the deepest leaf (after the rotations), one atomic per workgroup (1,500 per-wave atomics on one word took 15 of the
// kernel's 22 microseconds).
__global__ __launch_bounds__(256) void k_max_depth(int n, const int *parent, int *max_depth) {
    const int k = (int)(blockIdx.x * 256u + threadIdx.x);
    int depth = 0;
    if (k < n)
        for (int c = parent[(n - 1) + k]; c >= 0; c = parent[c]) depth++;
    for (int m = 32; m >= 1; m >>= 1) {
        const int o = __shfl_xor(depth, m);
        depth = o > depth ? o : depth;
    }
    __shared__ int red[4];
    if ((threadIdx.x & 63u) == 0u) red[threadIdx.x >> 6] = depth;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int a = red[0] > red[1] ? red[0] : red[1], b = red[2] > red[3] ? red[2] : red[3];
        atomicMax(max_depth, a > b ? a : b);
    }
}
// ---- subtree rebuild (the second quality pass; CPU statement and rationale: host/bvh.cpp, lbvh::rebuild_subtrees) ----
constexpr int kRebuildLeaves = 64;  // == GLRT_LBVH_REBUILD_LEAVES (glrt_host.h)
constexpr int kRebuildSahLevels = 20;  // as in host/bvh.cpp: deeper segments are halved
// The rebuild forms unions of many boxes in an order of its own (scans across lanes; the CPU statement: sequential sweeps).  min and
// max give the same result in any order once no operand is a negative zero, so the leaf boxes are read as x + 0.0f (-0 -> +0,
// everything else unchanged) and the unions use the plain instructions.
// Leaves below every internal node, capped at kRebuildLeaves + 1: one thread per node, a depth-first walk that stops at the cap.  The walk keeps no stack (round 4; a
// per-thread int[64] lived in scratch memory): it finds its way by the parent links -- from above to the left child, back from the left child to the right one, back
// from the right child up -- so the kernel needs no private memory, and no depth limit either (the CPU statement, host/bvh.cpp: rebuild_subtrees, counts exactly as well).
__global__ __launch_bounds__(256) void k_subtree_count(int n, const float *nodes, const int *parent, int *count) {
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n - 1) return;
    int v = i, prev = -1, cnt = 0;  // prev: the node the walk came from (-1: from above)
    for (;;) {
        if (v >= n - 1) {  // a leaf
            if (++cnt > kRebuildLeaves) break;
            prev = v; v = parent[v];
            continue;
        }
        const int l = (int)nodes[9 * (size_t)v + 6], r = (int)nodes[9 * (size_t)v + 7];
        if (prev == l) { prev = v; v = r; }
        else if (prev == r) { if (v == i) break; prev = v; v = parent[v]; }
        else { prev = v; v = l; }
    }
    count[i] = cnt < kRebuildLeaves + 1 ? cnt : kRebuildLeaves + 1;
}
// The roots of the MAXIMAL subtrees with at most kRebuildLeaves leaves, as a list (in no particular order: the subtrees are disjoint).
__global__ __launch_bounds__(256) void k_subtree_roots(int n, const int *parent, const int *count, int *roots, int *n_roots) {
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    bool is_root = false;
    if (i < n - 1 && count[i] <= kRebuildLeaves) {
        const int pr = parent[i];
        is_root = pr < 0 || count[pr] > kRebuildLeaves;
    }
    const unsigned long long mask = __ballot(is_root);
    if (mask == 0ull) return;
    const int lane = (int)(threadIdx.x & 63u);
    int base = 0;
    if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(n_roots, __popcll(mask));
    base = __shfl(base, __ffsll((long long)mask) - 1);
    if (is_root) roots[base + __popcll(mask & ((1ull << lane) - 1ull))] = i;
}
// One wave per listed subtree (a fixed grid walks the list: a launch with a block per internal node spent 0.25 ms dispatching blocks that left at once).
// Lane p is position p of the subtree's leaf order.  The tree is built level by level; the nodes of a level are the SEGMENTS of the
// order, all of them processed at once: per axis the lanes rank themselves inside their segment (a count over the segment),
// segmented prefix / suffix unions by shuffles, one cost per lane, the segment's minimum by a loop over the segment.
// EXPLICIT (round 5, sahl.hip.h): the subtrees are not read off an existing tree but given as lists -- root slot, number of leaves, where the leaves' sorted positions
// start in `members` (ascending), and the first of the count - 2 consecutive slots of the subtree's other inner nodes.
struct ExplicitSubtrees {
    const int *root, *count, *member_off, *extra_base, *members;
    int n;
};
template <bool EXPLICIT>
__global__ __launch_bounds__(192) void k_rebuild_subtrees(int n, float *nodes, int *parent, const int *roots, const int *n_roots, ExplicitSubtrees ex) {
    // Three waves per subtree, one per axis: each keeps the same per-lane state (segment, slot, leaf) and sweeps its own axis; the
    // three candidates meet in LDS and every wave takes the same decision in axis order.  A subtree is a chain of dependent steps
    // with two or three subtrees per SIMD at 100 k triangles, so the kernel's time IS that chain: three waves cut it to a third.
    const int n_int = n - 1, tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ int q[2 * kRebuildLeaves], qn, slots[kRebuildLeaves], leaves[kRebuildLeaves], ord[kRebuildLeaves];
    __shared__ int ref_l[kRebuildLeaves], ref_r[kRebuildLeaves], side[kRebuildLeaves], moved[3][kRebuildLeaves];
    __shared__ float box[kRebuildLeaves][6], cen[kRebuildLeaves][3], r_cost[3][kRebuildLeaves];
    __shared__ int r_k[3][kRebuildLeaves], r_imb[3][kRebuildLeaves];
  const int n_sub = EXPLICIT ? ex.n : *n_roots;
  for (int ri = (int)blockIdx.x; ri < n_sub; ri += (int)gridDim.x) {
    const int r = EXPLICIT ? ex.root[ri] : roots[ri];
    __syncthreads();  // the previous subtree's last reads of the shared arrays are done
    int total, m;
    if (EXPLICIT) {
        m = ex.count[ri];
        total = 2 * m - 1;
        if (tid < m) leaves[tid] = n_int + ex.members[ex.member_off[ri] + tid];
        if (tid < m - 2) slots[tid] = ex.extra_base[ri] + tid;
    } else {
    // ---- gather the subtree's nodes (breadth-first), then its internal slots (without r) and leaves in ascending index order
    if (tid == 0) { q[0] = r; qn = 1; }
    __syncthreads();
    for (int head = 0;;) {
        const int end = qn;
        __syncthreads();  // everyone has read the count before anyone appends
        if (head >= end) break;
        for (int i = head + tid; i < end; i += 192) {
            const int v = q[i];
            if (v < n_int) {
                const int at = atomicAdd(&qn, 2);
                q[at] = (int)nodes[9 * (size_t)v + 6];
                q[at + 1] = (int)nodes[9 * (size_t)v + 7];
            }
        }
        head = end;
        __syncthreads();
    }
    total = qn; m = (total + 1) / 2;  // m leaves, m - 1 internal nodes
    for (int i = tid; i < total; i += 192) {
        const int v = q[i];
        if (v == r) continue;
        const bool is_leaf = v >= n_int;
        int rank = 0;
#pragma unroll 8
        for (int j = 0; j < total; j++) {
            const int w = q[j];
            if (w != r && (w >= n_int) == is_leaf && w < v) rank++;
        }
        if (is_leaf) leaves[rank] = v; else slots[rank] = v;
    }
    }
    __syncthreads();
    if (tid < m) {
        const float *L = nodes + 9 * (size_t)leaves[lane];
        for (int k = 0; k < 6; k++) box[lane][k] = L[k] + 0.0f;
        for (int k = 0; k < 3; k++) cen[lane][k] = centre(L[k], L[3 + k]);
    }
    __syncthreads();
    int a = 0, b = lane < m ? m : 0, slot = r;  // this lane's segment [a, b) and the node slot it is building; b - a < 2: nothing to do
    if (lane >= m) { a = lane; b = lane; }
    // This wave's order: the leaves (ranks in `leaves`) sorted by (centre on the wave's axis, leaf rank), sorted ONCE by a bitonic network over the wave -- idle lanes
    // carry a key above every leaf's and stay where they are; leaf ranks are distinct, so the result is THE sorted order.  A segment's members stay in the segment's
    // lanes on every axis; a split moves them by a stable partition (below), which keeps each side sorted.
    const int axis = wv;
    int item = lane;
    {
        int s_idle = lane < m ? 0 : 1 + lane;
        float s_key = lane < m ? cen[lane][axis] : 0.f;
#pragma unroll
        for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
            for (int jj = kk >> 1; jj >= 1; jj >>= 1) {
                const int o_idle = __shfl_xor(s_idle, jj), o_item = __shfl_xor(item, jj);
                const float o_key = __shfl_xor(s_key, jj);
                const bool o_less = o_idle < s_idle || (o_idle == s_idle && (o_key < s_key || (o_key == s_key && o_item < item)));
                const bool want_less = ((lane & jj) == 0) == ((lane & kk) == 0);  // this lane keeps the smaller of the pair
                if (o_less == want_less) { s_idle = o_idle; s_key = o_key; item = o_item; }
            }
        }
    }
    int next_slot = 0;
    for (int level = 0;; level++) {
        const bool busy = b - a >= 2;
        if (!__any(busy)) break;
        float all_lo[3] = {0.f, 0.f, 0.f}, all_hi[3] = {0.f, 0.f, 0.f};
        {
            // segmented inclusive prefix and suffix unions of the boxes in sorted order
            float plo[3], phi[3], slo[3], shi[3];
            for (int k = 0; k < 3; k++) { plo[k] = slo[k] = busy ? box[item][k] : 0.f; phi[k] = shi[k] = busy ? box[item][3 + k] : 0.f; }
            for (int d = 1; d < 64; d <<= 1) {
                for (int k = 0; k < 3; k++) {
                    const float ul = __shfl_up(plo[k], d), uh = __shfl_up(phi[k], d), dl = __shfl_down(slo[k], d), dh = __shfl_down(shi[k], d);
                    if (busy && lane - d >= a) { plo[k] = __builtin_fminf(plo[k], ul); phi[k] = __builtin_fmaxf(phi[k], uh); }
                    if (busy && lane + d < b) { slo[k] = __builtin_fminf(slo[k], dl); shi[k] = __builtin_fmaxf(shi[k], dh); }
                }
            }
            // the split in front of this lane: [a, lane) | [lane, b)
            float cost = __builtin_inff();
            {
                float ql[3], qh[3];
                for (int k = 0; k < 3; k++) { ql[k] = __shfl_up(plo[k], 1); qh[k] = __shfl_up(phi[k], 1); }
                if (busy && lane > a) cost = half_area9(ql, qh) * (float)(lane - a) + half_area9(slo, shi) * (float)(b - lane);
            }
            if (lane == a) for (int k = 0; k < 3; k++) { all_lo[k] = slo[k]; all_hi[k] = shi[k]; }  // (the same box on every axis; wave 0 writes it)
            // the segment's best split on this axis: lowest cost, then most balanced, then first -- a suffix minimum over the
            // segment's lanes by shuffles, read back from the segment's first lane.  A NaN cost never qualifies (as `c < bc` is
            // false for it in the CPU statement's loop); an infinite one does, on the balance rule.
            const bool cand = busy && lane > a && level < kRebuildSahLevels && cost == cost;
            float bc = cand ? cost : __builtin_inff();
            int bi = cand ? abs(2 * (lane - a) - (b - a)) : 0x7fffffff, bk = cand ? lane - a : 0x7fffffff;  // (bk < 0x7fffffff: a split qualified)
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const float oc = __shfl_down(bc, d);
                const int oi = __shfl_down(bi, d), ok = __shfl_down(bk, d);
                if (busy && lane + d < b && ok != 0x7fffffff &&
                    (bk == 0x7fffffff || oc < bc || (oc == bc && (oi < bi || (oi == bi && ok < bk))))) { bc = oc; bi = oi; bk = ok; }
            }
            r_cost[axis][lane] = __shfl(bc, a); r_imb[axis][lane] = __shfl(bi, a); r_k[axis][lane] = __shfl(bk, a);
        }
        __syncthreads();
        // every wave takes the same decision: the axes in order; the first axis' order is taken even if no split qualifies (non-finite boxes, or past the SAH
        // levels; as the CPU statement does)
        float best = __builtin_inff();
        int best_k = 1, best_axis = 0, best_imb = b - a;
        if (busy)
            for (int ax = 0; ax < 3; ax++) {
                const float bc = r_cost[ax][lane];
                const int bk = r_k[ax][lane], bi = r_imb[ax][lane];
                if (bk != 0x7fffffff && (bc < best || (bc == best && bi < best_imb))) { best = bc; best_k = bk; best_imb = bi; best_axis = ax; }
            }
        if (level >= kRebuildSahLevels) best_k = (b - a) / 2;
        // the winning axis' order is the segment's order: its wave publishes it, and which side each leaf goes to
        if (busy && best_axis == axis) { ord[lane] = item; side[item] = lane < a + best_k ? 1 : 0; }
        const bool leader = busy && lane == a;
        const int n_l = best_k, n_r = (b - a) - best_k;
        const int need = leader ? (n_l >= 2 ? 1 : 0) + (n_r >= 2 ? 1 : 0) : 0;
        int incl = need;
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        const int level_total = __shfl(incl, 63);
        __syncthreads();  // ord, side complete
        if (leader && wv == 0) {
            const int at = next_slot + incl - need;
            const int rl = n_l >= 2 ? slots[at] : leaves[ord[a]];
            const int rr = n_r >= 2 ? slots[at + (n_l >= 2 ? 1 : 0)] : leaves[ord[b - 1]];
            ref_l[a] = rl; ref_r[a] = rr;
            float *N = nodes + 9 * (size_t)slot;
            for (int k = 0; k < 3; k++) { N[k] = all_lo[k]; N[3 + k] = all_hi[k]; }
            N[6] = (float)rl; N[7] = (float)rr; N[8] = -1.0f;
            parent[rl] = slot;
            parent[rr] = slot;
        }
        next_slot += level_total;
        // stable partition of this wave's order by side
        {
            const int left = busy ? side[item] : 0;
            int cnt = left;
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(cnt, d); if (busy && lane - d >= a) cnt += o; }
            const int before = cnt - left;  // left-going members in front of this one
            if (busy) moved[axis][left ? a + before : a + best_k + (lane - a - before)] = item;
        }
        __syncthreads();
        if (busy) {
            item = moved[axis][lane];
            if (lane < a + best_k) { slot = ref_l[a]; b = a + best_k; }
            else { slot = ref_r[a]; a = a + best_k; }
        }
    }
  }
}
constexpr int kRotationPasses = GLRT_LBVH_ROTATION_PASSES;  // == GLRT_LBVH_ROTATION_PASSES (glrt_host.h): the CPU statement must run the same sweeps

struct Workspace {
    void *p = nullptr;
    size_t bytes = 0;
};

// Device-side build.  d_vert / d_tri: wire-format buffers on the device; d_nodes: 9 * (2n - 1) floats.
// Returns hipSuccess or the failing call's error; *bad_index is set when a triangle references a vertex out of range.
inline hipError_t build(hipStream_t stream, const float *d_vert, unsigned n_vert, const float *d_tri, unsigned n, float *d_nodes,
                        Workspace &ws, int *max_depth_out, int *bad_index) {
#define LBVH_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)
    *bad_index = 0;
    *max_depth_out = 0;
    const size_t n_nodes = 2 * (size_t)n - 1;
    size_t sort_bytes = 0;
    LBVH_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (int)n,
                                               0, 64, stream));
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_keys0 = 0, o_keys1 = o_keys0 + up(8 * (size_t)n), o_parent = o_keys1 + up(8 * (size_t)n),
                 o_arrived = o_parent + up(4 * n_nodes), o_small = o_arrived + up(4 * (size_t)n), o_sort = o_small + 1024,
                 total = o_sort + up(sort_bytes);
    if (ws.bytes < total) {
        if (ws.p) (void)hipFree(ws.p);
        ws.p = nullptr; ws.bytes = 0;
        LBVH_TRY(hipMalloc(&ws.p, total));
        ws.bytes = total;
    }
    char *base = (char *)ws.p;
    unsigned long long *keys0 = (unsigned long long *)(base + o_keys0), *keys1 = (unsigned long long *)(base + o_keys1);
    int *parent = (int *)(base + o_parent);
    unsigned *arrived = (unsigned *)(base + o_arrived);
    unsigned *bounds = (unsigned *)(base + o_small);  // component a at word 32 * a (own 128-B line); [192] err, [193] max depth
    int *err = (int *)(bounds + 192), *max_depth = (int *)(bounds + 193);
    unsigned init[256] = {0};
    init[0] = init[32] = init[64] = 0xFFFFFFFFu;
    LBVH_TRY(hipMemcpyAsync(bounds, init, sizeof init, hipMemcpyHostToDevice, stream));
    LBVH_TRY(hipMemsetAsync(arrived, 0, 4 * (size_t)n, stream));
    const dim3 grid((n + 255) / 256), block(256);
    hipLaunchKernelGGL(k_centre_bounds, grid, block, 0, stream, d_vert, n_vert, d_tri, n, bounds, err);
    hipLaunchKernelGGL(k_keys, grid, block, 0, stream, d_vert, n_vert, d_tri, n, bounds, keys0);
    LBVH_TRY(hipGetLastError());
    LBVH_TRY(hipcub::DeviceRadixSort::SortKeys(base + o_sort, sort_bytes, keys0, keys1, (int)n, 0, 64, stream));
    if (n > 1) hipLaunchKernelGGL(k_hierarchy, grid, block, 0, stream, keys1, (int)n, d_nodes, parent);
    else LBVH_TRY(hipMemsetAsync(parent, 0xFF, 4, stream));  // the single leaf is the root: parent = -1
    hipLaunchKernelGGL(k_leaves, grid, block, 0, stream, d_vert, n_vert, d_tri, keys1, (int)n, d_nodes, parent, max_depth);
    LBVH_TRY(hipGetLastError());
    int out[2] = {0, 0};
    LBVH_TRY(hipMemcpyAsync(out, err, sizeof out, hipMemcpyDeviceToHost, stream));
    LBVH_TRY(hipStreamSynchronize(stream));
    *bad_index = out[0];
    *max_depth_out = out[1];
    if (out[0]) return hipSuccess;
    for (int round = 1; round <= out[1]; round++)
        hipLaunchKernelGGL(k_fit_round, grid, block, 0, stream, (int)n, round, d_nodes, (int *)arrived);
    LBVH_TRY(hipGetLastError());
    if (n >= 3) {  // quality pass: tree rotations, bottom-up by original depth, kRotationPasses sweeps
        int *level = (int *)arrived;  // the fit stamps are no longer needed
        int depth = out[1];           // deepest leaf; internal nodes are at depths 0 .. depth - 1
        for (int pass = 0; pass < kRotationPasses; pass++) {
            // levels as they are when the sweep starts (a rotation at an ancestor moves whole subtrees between nodes of one depth)
            hipLaunchKernelGGL(k_levels, grid, block, 0, stream, (int)n, parent, level);
            for (int d = depth - 1; d >= 0; d--)
                hipLaunchKernelGGL(k_rotate_level, grid, block, 0, stream, (int)n, d, level, d_nodes, parent);
            LBVH_TRY(hipMemsetAsync(max_depth, 0, sizeof(int), stream));
            hipLaunchKernelGGL(k_max_depth, grid, block, 0, stream, (int)n, parent, max_depth);
            LBVH_TRY(hipGetLastError());
            LBVH_TRY(hipMemcpyAsync(&depth, max_depth, sizeof(int), hipMemcpyDeviceToHost, stream));
            LBVH_TRY(hipStreamSynchronize(stream));  // the next sweep's launch count depends on it
        }
        // second quality pass: every maximal subtree of at most kRebuildLeaves leaves rebuilt with the exact sweep SAH
        int *count = (int *)arrived;  // the rotation levels are no longer needed
        int *roots = (int *)keys0, *n_roots = (int *)(bounds + 194);  // the unsorted keys are no longer needed
        LBVH_TRY(hipMemsetAsync(n_roots, 0, sizeof(int), stream));
        hipLaunchKernelGGL(k_subtree_count, grid, block, 0, stream, (int)n, d_nodes, parent, count);
        hipLaunchKernelGGL(k_subtree_roots, grid, block, 0, stream, (int)n, parent, count, roots, n_roots);
        hipLaunchKernelGGL(k_rebuild_subtrees<false>, dim3(std::min<unsigned>(n - 1, 8192u)), dim3(192), 0, stream, (int)n, d_nodes, parent, (const int *)roots, (const int *)n_roots, ExplicitSubtrees{});
        LBVH_TRY(hipMemsetAsync(max_depth, 0, sizeof(int), stream));
        hipLaunchKernelGGL(k_max_depth, grid, block, 0, stream, (int)n, parent, max_depth);
        LBVH_TRY(hipGetLastError());
        LBVH_TRY(hipMemcpyAsync(&depth, max_depth, sizeof(int), hipMemcpyDeviceToHost, stream));
        LBVH_TRY(hipStreamSynchronize(stream));
        *max_depth_out = depth;
    }
    LBVH_TRY(hipStreamSynchronize(stream));
    return hipSuccess;
#undef LBVH_TRY
}

}  // namespace lbvh
}  // namespace glrtx
