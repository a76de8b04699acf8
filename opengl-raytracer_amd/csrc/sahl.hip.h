// sahl.hip.h -- "SAH by levels": a binned-SAH BVH built top-down on the device, one tree level per round of launches, joined at the bottom to the exact-sweep
// builder of the LBVH pass (lbvh.hip.h: k_rebuild_subtrees).  SURVEY.md 8(f) f1; round 5.
//
// Why (profiles/r05_tree_study.txt): the Morton tree with rotations and 64-leaf rebuilds costs config 5 3.6 % more traversal steps than the CPU binned-SAH tree, and the
// gap only closes when subtrees of >= 8192 leaves are rebuilt -- it is the TOP of a Morton tree that is behind.  The algorithm, its tie-breaks and its node numbering
// are stated in host/bvh.cpp ("SAH by levels", glrt_bvh_build_sah_levels), which produces the same nodes bit for bit (tests/test_gpu_parity.py); in short:
//   triangles in Morton order (position k = leaf node n - 1 + k); every OPEN segment (> 64 members) is split per level by the best of 3 x 15 bin boundaries (16 bins over the
//   bounds of its members' centres; cost = area * count on both sides), or by position when no boundary qualifies / 40 levels deep; a child with 2 .. 64 members is CLOSED and
//   built by the exact sweep SAH from its leaves; nodes are numbered breadth-first, the closed subtrees' inner nodes behind them.
// On the device a level is seven launches: tables cleared; bounds of the segments' centres and positions; bin counts and boxes; one wave per segment choosing its split; a scan numbering
// the children; the members moving to their child.  All float reductions are min / max of values without negative zeros, formed with atomics on order-preserving unsigned
// keys (aggregated per block in LDS, a slot per segment the block meets: at the top of the tree every member of the scene lands in the same 350 words) -- the result does
// not depend on the order of the operands, so it equals the CPU statement's sequential loops.
#pragma once
#include "lbvh.hip.h"

namespace glrtx {
namespace sahl {
using lbvh::f2ord;
using lbvh::ord2f;

constexpr int kBins = 16;
constexpr int kClosed = lbvh::kRebuildLeaves;
constexpr int kDepthCap = 40;
// per open segment: [0, 48) counts (axis * 16 + bin), [48, 64) position-bin counts, [64, 64 + 288) boxes ((axis * 16 + bin) * 6 + {lo xyz, hi xyz}) as ordered keys
constexpr int kBinWords = 64 + 3 * kBins * 6;
struct Split {
    int axis, bin, nl, nr;  // axis 3: by position
    float lo, scale;
    int p0, prange;
};
struct Counters {
    int next_id, n_open_next, n_closed, pad;
};

__device__ __forceinline__ int bin_of(float c, float lo, float scale) {
    const float f = (c - lo) * scale;
    if (!(f >= 0.0f)) return 0;
    return f >= (float)kBins ? kBins - 1 : (int)f;
}
__device__ __forceinline__ int rank_bin(int k, int p0, int prange) {
    return (int)(((unsigned long long)(unsigned)(k - p0) * (unsigned long long)kBins) / (unsigned long long)(unsigned)prange);
}
// centre of the leaf at sorted position k (k_leaves wrote its box)
__device__ __forceinline__ void leaf_centre(const float *nodes, int n, int k, float c[3]) {
    const float *L = nodes + 9 * (size_t)((n - 1) + k);
    for (int a = 0; a < 3; a++) c[a] = lbvh::centre(L[a], L[3 + a]);
}

__global__ __launch_bounds__(256) void k_init(int n, int *seg, int *closed_of, int *parent, int first_open) {
    const int k = (int)(blockIdx.x * 256u + threadIdx.x);
    if (k < n) { seg[k] = first_open ? 0 : -1; closed_of[k] = first_open ? -1 : 0; }
    if (k < 2 * n - 1) parent[k] = -1;
}

// Layout of the per-segment tables: word w of segment s at [w * stride + s] (stride = the table's segment capacity): at the top of the tree -- one segment, every
// member of the scene -- the words then lie in different cache lines (atomics on one line serialise at ~90 per microsecond: with the words of a segment side by side
// the root level alone took 0.4 ms), and a thread per segment reads them coalesced.
// cb: 8 words -- [0..2] min / [3..5] max of the members' centres (ordered keys), [6] min / [7] max position
__device__ __forceinline__ void seg_tables_init(int S, int stride, unsigned *cb, unsigned *bins, long long first, long long step) {
    for (long long i = first; i < (long long)S * kBinWords; i += step) {
        const int w = (int)(i / S), s = (int)(i - (long long)w * S);
        if (w < 8) cb[(size_t)w * stride + s] = (w < 3 || w == 6) ? 0xFFFFFFFFu : 0u;
        bins[(size_t)w * stride + s] = w < 64 ? 0u : (((w - 64) % 6) < 3 ? 0xFFFFFFFFu : 0u);
    }
}
// the tables of the first level (the later levels' are initialised by the level before them: k_seg_partition)
__global__ __launch_bounds__(256) void k_seg_init(const int *S_ptr, int stride, unsigned *cb, unsigned *bins) {
    seg_tables_init(*S_ptr, stride, cb, bins, (long long)(blockIdx.x * 256u + threadIdx.x), (long long)gridDim.x * 256);
}
// A block's 256 consecutive positions belong to a handful of segments (the members are in Morton order: one segment near the top of the tree, three to six once the
// segments are down to a hundred members).  Each block therefore keeps up to kSlots segments' tables in LDS -- a thread claims a slot for its segment with a compare-and-swap
// on the slot's key -- forms the sums / minima / maxima there and sends one global atomic per non-empty word and slot; a thread that finds no slot uses global atomics itself.
// (Per-lane global atomics alone: 2.1 + 0.8 ms of a 4.5 ms build of 100 k triangles; aggregated only where a whole wave or block shares a segment: 1.5 + 0.9 ms.)
constexpr int kSlots = 8;
__device__ __forceinline__ int claim_slot(int s, int *slot_key) {
    for (int p = 0; p < kSlots; p++) {
        const int i = (s + p) & (kSlots - 1);
        const int old = atomicCAS(&slot_key[i], -1, s);
        if (old == -1 || old == s) return i;
    }
    return -1;
}
__global__ __launch_bounds__(256) void k_seg_bounds(int n, int stride, const float *nodes, const int *seg, unsigned *cb) {
    __shared__ int slot_key[kSlots];
    __shared__ unsigned tab[kSlots][8];
    const int k = (int)(blockIdx.x * 256u + threadIdx.x);
    const int s = k < n ? seg[k] : -1;
    const bool on = s >= 0;
    if (threadIdx.x < (unsigned)kSlots) slot_key[threadIdx.x] = -1;
    if (threadIdx.x < (unsigned)(kSlots * 8)) tab[threadIdx.x >> 3][threadIdx.x & 7u] = ((threadIdx.x & 7u) < 3u || (threadIdx.x & 7u) == 6u) ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    if (on) {
        float c[3];
        leaf_centre(nodes, n, k, c);
        unsigned v[8];
        for (int a = 0; a < 3; a++) v[a] = v[3 + a] = f2ord(c[a] + 0.0f);
        v[6] = v[7] = (unsigned)k;
        const int slot = claim_slot(s, slot_key);
        for (int w = 0; w < 8; w++) {
            unsigned *dst = slot >= 0 ? &tab[slot][w] : &cb[(size_t)w * stride + s];
            if (w < 3 || w == 6) atomicMin(dst, v[w]);
            else atomicMax(dst, v[w]);
        }
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)(kSlots * 8)) {
        const int slot = (int)(threadIdx.x >> 3), w = (int)(threadIdx.x & 7u), sk = slot_key[slot];
        if (sk >= 0) {
            if (w < 3 || w == 6) atomicMin(&cb[(size_t)w * stride + sk], tab[slot][w]);
            else atomicMax(&cb[(size_t)w * stride + sk], tab[slot][w]);
        }
    }
}

// bin counts and boxes of every open segment; words [0, 48) counts (axis * 16 + bin), [48, 64) position-bin counts, [64, ...) boxes ((axis * 16 + bin) * 6 + {lo, hi})
__global__ __launch_bounds__(256) void k_seg_bin(int n, int stride, const float *nodes, const int *seg, const unsigned *cb, unsigned *bins) {
    __shared__ int slot_key[kSlots];
    __shared__ unsigned tab[kSlots][kBinWords];
    const int k = (int)(blockIdx.x * 256u + threadIdx.x);
    const int s = k < n ? seg[k] : -1;
    const bool on = s >= 0;
    if (threadIdx.x < (unsigned)kSlots) slot_key[threadIdx.x] = -1;
    for (int i = (int)threadIdx.x; i < kSlots * kBinWords; i += 256) {
        const int w = i % kBinWords;
        (&tab[0][0])[i] = w < 64 ? 0u : (((w - 64) % 6) < 3 ? 0xFFFFFFFFu : 0u);
    }
    __syncthreads();
    if (on) {
        float c[3];
        leaf_centre(nodes, n, k, c);
        const float *L = nodes + 9 * (size_t)((n - 1) + k);
        unsigned bx[6];
        for (int a = 0; a < 3; a++) { bx[a] = f2ord(L[a] + 0.0f); bx[3 + a] = f2ord(L[3 + a] + 0.0f); }
        const int slot = claim_slot(s, slot_key);
        for (int a = 0; a < 3; a++) {
            const float lo = ord2f(cb[(size_t)a * stride + s]), ext = ord2f(cb[(size_t)(3 + a) * stride + s]) - lo;
            if (!(ext > 0.0f)) continue;
            const int e = a * kBins + bin_of(c[a], lo, (float)kBins / ext);
            if (slot >= 0) {
                atomicAdd(&tab[slot][e], 1u);
                for (int w = 0; w < 3; w++) { atomicMin(&tab[slot][64 + e * 6 + w], bx[w]); atomicMax(&tab[slot][64 + e * 6 + 3 + w], bx[3 + w]); }
            } else {
                atomicAdd(&bins[(size_t)e * stride + s], 1u);
                for (int w = 0; w < 3; w++) { atomicMin(&bins[(size_t)(64 + e * 6 + w) * stride + s], bx[w]); atomicMax(&bins[(size_t)(64 + e * 6 + 3 + w) * stride + s], bx[3 + w]); }
            }
        }
        const int p0 = (int)cb[(size_t)6 * stride + s];
        const int e = 48 + rank_bin(k, p0, (int)cb[(size_t)7 * stride + s] - p0 + 1);
        if (slot >= 0) atomicAdd(&tab[slot][e], 1u);
        else atomicAdd(&bins[(size_t)e * stride + s], 1u);
    }
    __syncthreads();
    for (int i = (int)threadIdx.x; i < kSlots * kBinWords; i += 256) {
        const int slot = i / kBinWords, w = i - slot * kBinWords, sk = slot_key[slot];
        if (sk < 0) continue;
        const unsigned x = tab[slot][w];
        if (w < 64) { if (x) atomicAdd(&bins[(size_t)w * stride + sk], x); }
        else {
            if (tab[slot][(w - 64) / 6] == 0u) continue;  // an empty bin
            if (((w - 64) % 6) < 3) atomicMin(&bins[(size_t)w * stride + sk], x);
            else atomicMax(&bins[(size_t)w * stride + sk], x);
        }
    }
}

__device__ __forceinline__ float half_area(const float *lo, const float *hi) { return lbvh::half_area9(lo, hi); }

// One WAVE per open segment: its split, and how many nodes / open / closed children it needs (host/bvh.cpp has the rules).  Lane = axis * 16 + bin for the three axes
// (lanes 0 .. 47), the 16 position bins in lanes 48 .. 63: prefix and suffix unions of the bin boxes by shuffles inside the groups of 16, one cost per bin boundary,
// the winner by a wave-wide minimum of (cost, lane) -- the first in (axis, bin) order among equal costs, as the CPU statement's sequential "strictly less" keeps it.
// (min / max of values without negative zeros: the unions do not depend on the order they are formed in.)
__global__ __launch_bounds__(64) void k_seg_split(const int *S_ptr, int stride, int level, const int *open_count, const unsigned *cb, const unsigned *bins, Split *split, int *need) {
    const int S = *S_ptr;
    const int s = (int)blockIdx.x, lane = (int)threadIdx.x;
    if (s >= S) return;
    const int grp = lane >> 4, q = lane & 15;  // grp 0..2: axis, 3: position
    const int count = open_count[s];
    const float inf = __builtin_inff();
    const int e = lane;  // count word: [0, 48) axis bins, [48, 64) position bins
    const int cnt = (int)bins[(size_t)e * stride + s];
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    if (grp < 3 && cnt > 0)
        for (int w = 0; w < 3; w++) { lo[w] = ord2f(bins[(size_t)(64 + e * 6 + w) * stride + s]); hi[w] = ord2f(bins[(size_t)(64 + e * 6 + 3 + w) * stride + s]); }
    // inclusive prefix (bins 0 .. q) and suffix (bins q .. 15) inside the group of 16
    float plo[3], phi[3], slo[3], shi[3];
    int pc = cnt, sc = cnt;
    for (int w = 0; w < 3; w++) { plo[w] = slo[w] = lo[w]; phi[w] = shi[w] = hi[w]; }
    for (int d = 1; d < 16; d <<= 1) {
        const int upc = __shfl_up(pc, d, 16), dnc = __shfl_down(sc, d, 16);
        if (q >= d) pc += upc;
        if (q + d < 16) sc += dnc;
        for (int w = 0; w < 3; w++) {
            const float ul = __shfl_up(plo[w], d, 16), uh = __shfl_up(phi[w], d, 16), dl = __shfl_down(slo[w], d, 16), dh = __shfl_down(shi[w], d, 16);
            if (q >= d) { plo[w] = __builtin_fminf(plo[w], ul); phi[w] = __builtin_fmaxf(phi[w], uh); }
            if (q + d < 16) { slo[w] = __builtin_fminf(slo[w], dl); shi[w] = __builtin_fmaxf(shi[w], dh); }
        }
    }
    // the boundary behind bin q: left = bins 0 .. q, right = bins q + 1 .. 15
    const int rc = __shfl_down(sc, 1, 16);
    float rlo[3], rhi[3];
    for (int w = 0; w < 3; w++) { rlo[w] = __shfl_down(slo[w], 1, 16); rhi[w] = __shfl_down(shi[w], 1, 16); }
    float ext = 0.0f;
    if (grp < 3) ext = ord2f(cb[(size_t)(3 + grp) * stride + s]) - ord2f(cb[(size_t)grp * stride + s]);
    float cost = inf;
    if (grp < 3 && q < 15 && level < kDepthCap && ext > 0.0f && pc > 0 && rc > 0) cost = half_area(plo, phi) * (float)pc + half_area(rlo, rhi) * (float)rc;
    // wave-wide minimum of (cost, lane): a NaN cost (boxes at infinity) never wins, as in the sequential `cost < best`
    float bc = cost < inf ? cost : inf;
    int bl = cost < inf ? lane : 64;
    for (int d = 32; d >= 1; d >>= 1) {
        const float oc = __shfl_xor(bc, d);
        const int ol = __shfl_xor(bl, d);
        if (oc < bc || (oc == bc && ol < bl)) { bc = oc; bl = ol; }
    }
    // the split by position: the most even boundary with members on both sides, the first among equals
    int imb = 0x7fffffff, il = 64;
    if (grp == 3 && q < 15 && pc > 0 && pc < count) { imb = abs(2 * pc - count); il = lane; }
    for (int d = 32; d >= 1; d >>= 1) {
        const int oi = __shfl_xor(imb, d), ol = __shfl_xor(il, d);
        if (oi < imb || (oi == imb && ol < il)) { imb = oi; il = ol; }
    }
    const int win = bl < 64 ? bl : il;           // (il < 64 always: positions are distinct, so some boundary separates them)
    const int nl = __shfl(pc, win < 64 ? win : 0);
    if (lane != 0) return;
    Split sp;
    sp.p0 = (int)cb[(size_t)6 * stride + s]; sp.prange = (int)cb[(size_t)7 * stride + s] - sp.p0 + 1;
    if (bl < 64) {
        const int a = bl >> 4;
        sp.axis = a; sp.bin = bl & 15; sp.nl = nl; sp.nr = count - nl;
        sp.lo = ord2f(cb[(size_t)a * stride + s]); sp.scale = (float)kBins / (ord2f(cb[(size_t)(3 + a) * stride + s]) - sp.lo);
    } else {
        sp.axis = 3; sp.bin = il < 64 ? (il & 15) : -1; sp.nl = il < 64 ? nl : 0; sp.nr = count - sp.nl; sp.lo = 0.0f; sp.scale = 0.0f;
    }
    split[s] = sp;
    // need[s * 4 + {0: nodes, 1: open children, 2: closed children}]
    int nn = 0, no = 0, nc = 0;
    for (int side = 0; side < 2; side++) {
        const int c = side ? sp.nr : sp.nl;
        if (c >= 2) nn++;
        if (c > kClosed) no++;
        else if (c >= 2) nc++;
    }
    need[s * 4 + 0] = nn; need[s * 4 + 1] = no; need[s * 4 + 2] = nc; need[s * 4 + 3] = 0;
}

// exclusive prefix sums over the 1024 per-thread partial sums of a one-workgroup scan, W columns at once: shuffles inside each of the 16 waves, then over the waves'
// totals.  part[t][w]: in the thread's sum, out the sum of the threads before it + start[w]; returns the grand total + start[w] in total[w] (every thread).
template <int W>
__device__ __forceinline__ void block_exclusive_1024(int (*part)[W], int (*wave_tot)[W], const int *start, int *total) {
    const int t = (int)threadIdx.x, lane = t & 63, wv = t >> 6;
    int v[W], incl[W];
    for (int w = 0; w < W; w++) { v[w] = part[t][w]; incl[w] = v[w]; }
    for (int d = 1; d < 64; d <<= 1)
        for (int w = 0; w < W; w++) { const int o = __shfl_up(incl[w], d); if (lane >= d) incl[w] += o; }
    if (lane == 63) for (int w = 0; w < W; w++) wave_tot[wv][w] = incl[w];
    __syncthreads();
    for (int w = 0; w < W; w++) {
        int before = start[w], all = start[w];
        for (int i = 0; i < 16; i++) { const int x = wave_tot[i][w]; all += x; if (i < wv) before += x; }
        part[t][w] = before + incl[w] - v[w];
        total[w] = all;
    }
    __syncthreads();
}

// exclusive prefix sums of need[] (three interleaved columns) in segment order, ONE workgroup; the running totals continue the counters
__global__ __launch_bounds__(1024) void k_seg_scan(const int *S_ptr, int *need, Counters *cnt, int *S_next, int *level_first_next) {
    const int S = *S_ptr;
    __shared__ int part[1024][3], wave_tot[16][3];
    const int t = (int)threadIdx.x;
    const int per = (S + 1023) / 1024;
    const int a = t * per, b = a + per < S ? a + per : S;
    int sum[3] = {0, 0, 0};
    for (int s = a; s < b; s++)
        for (int w = 0; w < 3; w++) sum[w] += need[s * 4 + w];
    for (int w = 0; w < 3; w++) part[t][w] = sum[w];
    const int start[3] = {cnt->next_id, 0, cnt->n_closed};
    int base[3];
    __syncthreads();  // (the counters are read before thread 0 writes them below)
    block_exclusive_1024<3>(part, wave_tot, start, base);
    int run[3] = {part[t][0], part[t][1], part[t][2]};
    for (int s = a; s < b; s++)
        for (int w = 0; w < 3; w++) { const int v = need[s * 4 + w]; need[s * 4 + w] = run[w]; run[w] += v; }
    if (t == 0) { cnt->next_id = base[0]; cnt->n_open_next = base[1]; cnt->n_closed = base[2]; *S_next = base[1]; *level_first_next = base[0]; }
}

// one thread per open segment: number its children, write its node's child links, the next level's open list and the closed list
__global__ __launch_bounds__(64) void k_seg_assign(const int *S_ptr, int n, const int *open_node, const Split *split, const int *need, float *nodes, int *parent, int *next_node,
                                                   int *next_count, int *closed_root, int *closed_count, int *child_info) {
    const int S = *S_ptr;
    const int s = (int)(blockIdx.x * 64u + threadIdx.x);
    if (s >= S) return;
    const Split sp = split[s];
    int id = need[s * 4 + 0], io = need[s * 4 + 1], ic = need[s * 4 + 2];
    const int node = open_node[s];
    float *N = nodes + 9 * (size_t)node;
    for (int side = 0; side < 2; side++) {
        const int c = side ? sp.nr : sp.nl;
        int ref = -1, oi = -1, ci = -1;
        if (c >= 2) {
            ref = id++;
            parent[ref] = node;
            if (c > kClosed) { oi = io++; next_node[oi] = ref; next_count[oi] = c; }
            else { ci = ic++; closed_root[ci] = ref; closed_count[ci] = c; }
            N[6 + side] = (float)ref;
        }
        child_info[s * 4 + side * 2 + 0] = oi;
        child_info[s * 4 + side * 2 + 1] = ci;
    }
    N[8] = -1.0f;
    (void)n;
}

// every member moves to its child (or becomes its parent's leaf child)
// (and, nobody reading this level's tables any more, initialises them for the next level's segments: a launch saved per level)
__global__ __launch_bounds__(256) void k_seg_partition(int n, const float *nodes_c, float *nodes, int *seg, int *closed_of, const int *open_node, const Split *split,
                                                       const int *child_info, int *parent, const int *S_next_ptr, int stride, unsigned *cb, unsigned *bins) {
    const int k = (int)(blockIdx.x * 256u + threadIdx.x);
    seg_tables_init(*S_next_ptr, stride, cb, bins, (long long)k, (long long)gridDim.x * 256);
    if (k >= n) return;
    const int s = seg[k];
    if (s < 0) return;
    const Split sp = split[s];
    int q;
    if (sp.axis == 3) q = rank_bin(k, sp.p0, sp.prange);
    else {
        float c[3];
        leaf_centre(nodes_c, n, k, c);
        q = bin_of(sp.axis == 0 ? c[0] : (sp.axis == 1 ? c[1] : c[2]), sp.lo, sp.scale);
    }
    const int side = q <= sp.bin ? 0 : 1;
    const int cnt = side ? sp.nr : sp.nl;
    if (cnt == 1) {
        const int node = open_node[s];
        nodes[9 * (size_t)node + 6 + side] = (float)((n - 1) + k);
        parent[(n - 1) + k] = node;
        seg[k] = -1;
    } else {
        const int oi = child_info[s * 4 + side * 2 + 0], ci = child_info[s * 4 + side * 2 + 1];
        if (oi >= 0) seg[k] = oi;
        else { seg[k] = -1; closed_of[k] = ci; }
    }
}

// closed subtrees: where their members start in the sorted list and where their inner nodes go; ONE workgroup (a chunked scan like k_seg_scan)
__global__ __launch_bounds__(1024) void k_closed_scan(int C, const int *closed_count, int first_extra, int *member_off, int *extra_base) {
    __shared__ int part[1024][2], wave_tot[16][2];
    const int t = (int)threadIdx.x;
    const int per = (C + 1023) / 1024;
    const int a = t * per, b = a + per < C ? a + per : C;
    int s0 = 0, s1 = 0;
    for (int j = a; j < b; j++) { s0 += closed_count[j]; s1 += closed_count[j] - 2; }
    part[t][0] = s0; part[t][1] = s1;
    const int start[2] = {0, first_extra};
    int total[2];
    __syncthreads();
    block_exclusive_1024<2>(part, wave_tot, start, total);
    int r0 = part[t][0], r1 = part[t][1];
    for (int j = a; j < b; j++) { member_off[j] = r0; extra_base[j] = r1; r0 += closed_count[j]; r1 += closed_count[j] - 2; }
}
__global__ __launch_bounds__(256) void k_closed_keys(int n, const int *closed_of, unsigned *key, unsigned *val) {
    const int k = (int)(blockIdx.x * 256u + threadIdx.x);
    if (k >= n) return;
    key[k] = closed_of[k] < 0 ? 0xFFFFFFFFu : (unsigned)closed_of[k];
    val[k] = (unsigned)k;
}

// boxes of the nodes above the closed subtrees, one launch per level from the deepest up: node ids [i0, i1) were numbered in one level; their children are leaves,
// closed roots (built) or nodes of a deeper level (fitted by an earlier launch)
__global__ __launch_bounds__(256) void k_top_fit(int i0, int i1, float *nodes, const unsigned char *is_closed_root) {
    const int i = i0 + (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= i1 || is_closed_root[i]) return;
    float *N = nodes + 9 * (size_t)i;
    const float *A = nodes + 9 * (size_t)(int)N[6], *B = nodes + 9 * (size_t)(int)N[7];
    for (int w = 0; w < 3; w++) {
        N[w] = __builtin_fminf(A[w] + 0.0f, B[w] + 0.0f);
        N[3 + w] = __builtin_fmaxf(A[3 + w] + 0.0f, B[3 + w] + 0.0f);
    }
}
// the same for the levels [0, L_top] in ONE workgroup, deepest first (a level near the root has a handful of nodes: a launch each was mostly launch)
__global__ __launch_bounds__(1024) void k_top_fit_levels(int L_top, const int *level_first, float *nodes, const unsigned char *is_closed_root) {
    for (int L = L_top; L >= 0; L--) {
        const int i0 = level_first[L], i1 = level_first[L + 1];
        for (int i = i0 + (int)threadIdx.x; i < i1; i += 1024) {
            if (is_closed_root[i]) continue;
            float *N = nodes + 9 * (size_t)i;
            const float *A = nodes + 9 * (size_t)(int)N[6], *B = nodes + 9 * (size_t)(int)N[7];
            for (int w = 0; w < 3; w++) {
                N[w] = __builtin_fminf(A[w] + 0.0f, B[w] + 0.0f);
                N[3 + w] = __builtin_fmaxf(A[3 + w] + 0.0f, B[3 + w] + 0.0f);
            }
        }
        __threadfence_block();
        __syncthreads();  // the level's boxes are written before the level above reads them
    }
}
__global__ __launch_bounds__(256) void k_mark_closed(int C, const int *closed_root, unsigned char *is_closed_root) {
    const int j = (int)(blockIdx.x * 256u + threadIdx.x);
    if (j < C) is_closed_root[closed_root[j]] = 1;
}

// Device-side build.  Same contract as lbvh::build.  *build_levels: the number of levels the top-down phase ran.
inline hipError_t build(hipStream_t stream, const float *d_vert, unsigned n_vert, const float *d_tri, unsigned n, float *d_nodes, lbvh::Workspace &ws, int *max_depth_out,
                        int *bad_index, int *build_levels) {
#define SAHL_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)
    *bad_index = 0;
    *max_depth_out = 0;
    if (build_levels) *build_levels = 0;
    const size_t n_nodes = 2 * (size_t)n - 1;
    const int ni = (int)n;
    size_t sort_bytes = 0, sort2_bytes = 0;
    SAHL_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, ni, 0, 64, stream));
    SAHL_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort2_bytes, (const unsigned *)nullptr, (unsigned *)nullptr, (const unsigned *)nullptr, (unsigned *)nullptr, ni, 0, 32, stream));
    if (sort2_bytes > sort_bytes) sort_bytes = sort2_bytes;
    const size_t max_open = (size_t)n / (kClosed + 1) + 2, max_closed = (size_t)n / 2 + 2;
    const size_t stride_sz = (max_open + 31) / 32 * 32;  // segments per word of the cb / bins tables (a multiple of a cache line)
    const int stride = (int)stride_sz;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += up(bytes); return o; };
    const size_t o_keys0 = take(8 * (size_t)n), o_keys1 = take(8 * (size_t)n), o_parent = take(4 * n_nodes), o_seg = take(4 * (size_t)n), o_closed_of = take(4 * (size_t)n),
                 o_small = take(1024), o_cb = take(32 * stride_sz), o_bins = take(4 * (size_t)kBinWords * stride_sz), o_split = take(sizeof(Split) * max_open),
                 o_need = take(16 * max_open), o_info = take(16 * max_open), o_open = take(4 * 4 * max_open), o_croot = take(4 * max_closed), o_ccount = take(4 * max_closed),
                 o_moff = take(4 * max_closed), o_extra = take(4 * max_closed), o_flag = take(n_nodes), o_lvl = take(4 * 2 * (256 + 2)), o_sort = take(sort_bytes);
    const size_t total = off;
    if (ws.bytes < total) {
        if (ws.p) (void)hipFree(ws.p);
        ws.p = nullptr; ws.bytes = 0;
        SAHL_TRY(hipMalloc(&ws.p, total));
        ws.bytes = total;
    }
    char *base = (char *)ws.p;
    unsigned long long *keys0 = (unsigned long long *)(base + o_keys0), *keys1 = (unsigned long long *)(base + o_keys1);
    int *parent = (int *)(base + o_parent), *seg = (int *)(base + o_seg), *closed_of = (int *)(base + o_closed_of);
    unsigned *bounds = (unsigned *)(base + o_small);  // as lbvh::build: component a at word 32 * a; [192] err, [193] max depth; [200..203] Counters
    int *err = (int *)(bounds + 192), *max_depth = (int *)(bounds + 193);
    Counters *cnt = (Counters *)(bounds + 200);
    unsigned *cb = (unsigned *)(base + o_cb), *bins = (unsigned *)(base + o_bins);
    Split *split = (Split *)(base + o_split);
    int *need = (int *)(base + o_need), *child_info = (int *)(base + o_info);
    int *open_node[2] = {(int *)(base + o_open), (int *)(base + o_open) + 2 * max_open}, *open_count[2] = {(int *)(base + o_open) + max_open, (int *)(base + o_open) + 3 * max_open};
    int *closed_root = (int *)(base + o_croot), *closed_count = (int *)(base + o_ccount), *member_off = (int *)(base + o_moff), *extra_base = (int *)(base + o_extra);
    unsigned char *is_closed_root = (unsigned char *)(base + o_flag);
    unsigned init[256] = {0};
    init[0] = init[32] = init[64] = 0xFFFFFFFFu;
    init[200] = 1u;  // next_id: node 0 is the root segment's
    SAHL_TRY(hipMemcpyAsync(bounds, init, sizeof init, hipMemcpyHostToDevice, stream));
    const dim3 grid((n + 255) / 256), block(256);
    hipLaunchKernelGGL(lbvh::k_centre_bounds, grid, block, 0, stream, d_vert, n_vert, d_tri, n, bounds, err);
    hipLaunchKernelGGL(lbvh::k_keys, grid, block, 0, stream, d_vert, n_vert, d_tri, n, bounds, keys0);
    SAHL_TRY(hipGetLastError());
    SAHL_TRY(hipcub::DeviceRadixSort::SortKeys(base + o_sort, sort_bytes, keys0, keys1, ni, 0, 64, stream));
    const bool first_open = ni > kClosed;
    hipLaunchKernelGGL(k_init, dim3((unsigned)((n_nodes + 255) / 256)), block, 0, stream, ni, seg, closed_of, parent, first_open ? 1 : 0);
    hipLaunchKernelGGL(lbvh::k_leaves, grid, block, 0, stream, d_vert, n_vert, d_tri, keys1, ni, d_nodes, parent, max_depth);
    SAHL_TRY(hipGetLastError());
    int bad = 0;
    SAHL_TRY(hipMemcpyAsync(&bad, err, sizeof bad, hipMemcpyDeviceToHost, stream));
    SAHL_TRY(hipStreamSynchronize(stream));
    *bad_index = bad;
    if (bad) return hipSuccess;
    if (n == 1) return hipSuccess;  // the single leaf is the root (k_leaves wrote it at node 0)
    // ---- top-down phase.  The host does not learn a level's segment count before it launches the next one: the per-segment kernels are launched over the tables'
    // capacity and read the count from the device (S_of[level], written by the previous level's scan), and the host looks only every few levels whether segments are
    // still open (a read-back per level cost a quarter of the build).  Node ranges per level (level_first) are read back once, at the end.
    constexpr int kMaxLevels = 256;
    int *S_of = (int *)(base + o_lvl), *level_first_dev = S_of + kMaxLevels + 2;  // S_of[L]: open segments of level L; level_first_dev[L + 1]: the running node count after level L - 1
    Counters h{1, 0, 0, 0};
    {
        std::vector<int> lv(2 * (kMaxLevels + 2), 0);
        lv[0] = first_open ? 1 : 0;
        lv[(size_t)kMaxLevels + 2 + 1] = 1;  // level_first_dev[1] = 1: ids [0, 1) are the root's
        SAHL_TRY(hipMemcpyAsync(S_of, lv.data(), lv.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    }
    if (first_open) {
        const int root_open[2] = {0, ni};
        SAHL_TRY(hipMemcpyAsync(open_node[0], &root_open[0], 4, hipMemcpyHostToDevice, stream));
        SAHL_TRY(hipMemcpyAsync(open_count[0], &root_open[1], 4, hipMemcpyHostToDevice, stream));
    } else {
        const int c0[2] = {0, ni};
        SAHL_TRY(hipMemcpyAsync(closed_root, &c0[0], 4, hipMemcpyHostToDevice, stream));
        SAHL_TRY(hipMemcpyAsync(closed_count, &c0[1], 4, hipMemcpyHostToDevice, stream));
        h.n_closed = 1;
        SAHL_TRY(hipMemcpyAsync(cnt, &h, sizeof h, hipMemcpyHostToDevice, stream));
    }
    int level = 0;
    if (first_open) {
        // a balanced tree needs ceil(log2(n / 64)) levels; look for the first time there, then every other level
        int first_look = 1;
        while (((size_t)kClosed << first_look) < (size_t)n) first_look++;
        const unsigned cap = (unsigned)max_open;
        for (int look = first_look;; level++) {
            if (level >= kMaxLevels) return hipErrorUnknown;  // (cannot happen: from 40 levels on every split is by position and halves its segment)
            const int cur = level & 1;
            const int *Sp = S_of + level;
            if (level == 0) hipLaunchKernelGGL(k_seg_init, grid, block, 0, stream, Sp, stride, cb, bins);
            hipLaunchKernelGGL(k_seg_bounds, grid, block, 0, stream, ni, stride, (const float *)d_nodes, (const int *)seg, cb);
            hipLaunchKernelGGL(k_seg_bin, grid, block, 0, stream, ni, stride, (const float *)d_nodes, (const int *)seg, (const unsigned *)cb, bins);
            hipLaunchKernelGGL(k_seg_split, dim3(cap), dim3(64), 0, stream, Sp, stride, level, (const int *)open_count[cur], (const unsigned *)cb, (const unsigned *)bins, split, need);
            hipLaunchKernelGGL(k_seg_scan, dim3(1), dim3(1024), 0, stream, Sp, need, cnt, S_of + level + 1, level_first_dev + level + 2);
            hipLaunchKernelGGL(k_seg_assign, dim3((cap + 63u) / 64u), dim3(64), 0, stream, Sp, ni, (const int *)open_node[cur], (const Split *)split, (const int *)need, d_nodes, parent,
                               open_node[cur ^ 1], open_count[cur ^ 1], closed_root, closed_count, child_info);
            hipLaunchKernelGGL(k_seg_partition, grid, block, 0, stream, ni, (const float *)d_nodes, d_nodes, seg, closed_of, (const int *)open_node[cur], (const Split *)split,
                               (const int *)child_info, parent, (const int *)(S_of + level + 1), stride, cb, bins);
            SAHL_TRY(hipGetLastError());
            if (level + 1 >= look) {
                int open_next = 0;
                SAHL_TRY(hipMemcpyAsync(&open_next, S_of + level + 1, sizeof(int), hipMemcpyDeviceToHost, stream));
                SAHL_TRY(hipStreamSynchronize(stream));
                if (open_next == 0) { level++; break; }
                look = level + 1 + 2;
            }
        }
    }
    std::vector<int> level_first((size_t)level + 2);
    SAHL_TRY(hipMemcpyAsync(level_first.data(), level_first_dev, level_first.size() * sizeof(int), hipMemcpyDeviceToHost, stream));
    SAHL_TRY(hipMemcpyAsync(&h, cnt, sizeof h, hipMemcpyDeviceToHost, stream));
    SAHL_TRY(hipStreamSynchronize(stream));
    if (build_levels) *build_levels = level;
    // ---- closed subtrees: members gathered (stable sort by subtree), inner nodes numbered, exact sweep SAH
    const int C = h.n_closed, top_nodes = h.next_id;
    if (C > 0) {
        unsigned *key_in = (unsigned *)keys0, *val_in = key_in + n, *key_out = (unsigned *)keys1, *val_out = key_out + n;  // (the Morton keys are no longer needed: the leaves hold the triangle ids)
        hipLaunchKernelGGL(k_closed_keys, grid, block, 0, stream, ni, (const int *)closed_of, key_in, val_in);
        SAHL_TRY(hipcub::DeviceRadixSort::SortPairs(base + o_sort, sort_bytes, (const unsigned *)key_in, key_out, (const unsigned *)val_in, val_out, ni, 0, 32, stream));
        hipLaunchKernelGGL(k_closed_scan, dim3(1), dim3(1024), 0, stream, C, (const int *)closed_count, top_nodes, member_off, extra_base);
        SAHL_TRY(hipMemsetAsync(is_closed_root, 0, n_nodes, stream));
        hipLaunchKernelGGL(k_mark_closed, dim3((C + 255) / 256), block, 0, stream, C, (const int *)closed_root, is_closed_root);
        lbvh::ExplicitSubtrees ex{closed_root, closed_count, member_off, extra_base, (const int *)val_out, C};
        hipLaunchKernelGGL(lbvh::k_rebuild_subtrees<true>, dim3((unsigned)std::min(C, 8192)), dim3(192), 0, stream, ni, d_nodes, parent, (const int *)nullptr, (const int *)nullptr, ex);
        SAHL_TRY(hipGetLastError());
    }
    // boxes above the closed subtrees, deepest level first: wide levels a launch each, then the narrow ones at the top (<= 4096 nodes each) in one workgroup
    {
        int L = (int)level_first.size() - 2;
        for (; L >= 0; L--) {
            const int i0 = level_first[(size_t)L], i1 = level_first[(size_t)L + 1];
            if (i1 - i0 <= 4096) {
                bool narrow_above = true;  // (levels only widen downwards in practice; if one above is wide after all, keep launching per level)
                for (int M = L - 1; M >= 0; M--) narrow_above = narrow_above && level_first[(size_t)M + 1] - level_first[(size_t)M] <= 4096;
                if (narrow_above) break;
            }
            if (i1 > i0) hipLaunchKernelGGL(k_top_fit, dim3((unsigned)(i1 - i0 + 255) / 256), block, 0, stream, i0, i1, d_nodes, (const unsigned char *)is_closed_root);
        }
        if (L >= 0) hipLaunchKernelGGL(k_top_fit_levels, dim3(1), dim3(1024), 0, stream, L, (const int *)level_first_dev, d_nodes, (const unsigned char *)is_closed_root);
    }
    SAHL_TRY(hipMemsetAsync(max_depth, 0, sizeof(int), stream));
    hipLaunchKernelGGL(lbvh::k_max_depth, grid, block, 0, stream, ni, (const int *)parent, max_depth);
    SAHL_TRY(hipGetLastError());
    int depth = 0;
    SAHL_TRY(hipMemcpyAsync(&depth, max_depth, sizeof(int), hipMemcpyDeviceToHost, stream));
    SAHL_TRY(hipStreamSynchronize(stream));
    *max_depth_out = depth;
    return hipSuccess;
#undef SAHL_TRY
}

}  // namespace sahl
}  // namespace glrtx
