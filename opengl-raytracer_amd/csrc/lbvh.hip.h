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
    int i = (int)q;
    if (i < 0) i = 0;
    if (i > 1023) i = 1023;
    return (unsigned)i;
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
            for (int a = 0; a < 3; a++) lo[a] = hi[a] = f2ord(0.5f * (b.lo[a] + b.hi[a]));
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
        const float c = 0.5f * (b.lo[a] + b.hi[a]);
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
// Depth of the deepest leaf (after the rotations), one atomic per wave.
__global__ __launch_bounds__(256) void k_max_depth(int n, const int *parent, int *max_depth) {
    const int k = (int)(blockIdx.x * 256u + threadIdx.x);
    int depth = 0;
    if (k < n)
        for (int c = parent[(n - 1) + k]; c >= 0; c = parent[c]) depth++;
    for (int m = 32; m >= 1; m >>= 1) {
        const int o = __shfl_xor(depth, m);
        depth = o > depth ? o : depth;
    }
    if ((threadIdx.x & 63u) == 0u) atomicMax(max_depth, depth);
}
constexpr int kRotationPasses = 5;  // == GLRT_LBVH_ROTATION_PASSES (glrt_host.h): the CPU statement must run the same sweeps

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
        *max_depth_out = depth;
    }
    LBVH_TRY(hipStreamSynchronize(stream));
    return hipSuccess;
#undef LBVH_TRY
}

}  // namespace lbvh
}  // namespace glrtx
