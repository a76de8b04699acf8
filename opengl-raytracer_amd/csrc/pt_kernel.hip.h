// pt_kernel.hip.h -- the per-pixel path-tracing kernel for gfx950 (CDNA4, wave64).
//
// One work-item per pixel sample loop; this is the device-side replacement of the reference's
// fullscreen-quad fragment shader (src/shaders/raytrace.frag, launched by glDrawArrays at
// src/core/window.cpp:290).  Reference functions covered (raytrace.frag line ranges):
//   rand :104-111, main :565-614, radiance :409-559, intersect(Ray,Intersection) :276-335,
//   intersectBBox :259-274, intersect(Ray,Triangle) :226-257, sampleDirect :337-403,
//   fresnelConductor :158-178, GGX :180-184, microfacetGGXBRDF :186-193,
//   sampleGGXVNDF :195-214, weightedGGXPDF :216-219.
//
// Numerics contract: the reference image is a chaotic function of float32
// rounding, so every expression keeps the association order the reference's GL implementation
// evaluates (no contraction: this file is compiled with -ffp-contract=off; IEEE divide and sqrt;
// 3-component dots summed z,y,x; min/max returning the non-NaN operand) and sin/cos are the
// Cephes single-precision routines with explicit fused multiply-adds.  Nothing here uses
// v_rcp/v_rsq/v_sin approximations.
//
// Device data layout (built by glrtx_upload_scene from the reference wire format):
//   forks  : 4 x float4 (64 B) per interior BVH node, holding its CHILDREN's boxes:
//            {minL.xyz, refL} {maxL.xyz, refR} {minR.xyz, -} {maxR.xyz, -}
//            ref >= 0 -> fork index, ref < 0 -> ~triangle (leaf nodes are folded into their
//            parent's ref: the reference never tests a leaf's own box, raytrace.frag:310-331).
//            A child the wire format leaves out (children.x/y < 0; no builder of this repository does) is record -1 (triangle id 0):
//            zeros, "tested" like a leaf and never hit (det = 0), so that the step carries no test
//            for absent children.  The root's own box is in DevScene.
//   tris   : 4 x float4 (64 B, the shape of a fork record) per LEAF of the tree {v0.xyz, materialId} {v1-v0, next} {v2-v0, -} {-},
//            in the same array as the forks: the leaf with triangle id k (k >= 1, numbered in the order the traversal meets the leaves;
//            a hit carries this id, not the wire triangle index) at record index ~k; id 0 is the never-hit record.  `next` = REF_FIN, or
//            the ref of the triangle record chained behind this one: the two leaves of a fork with two leaf children are tested one
//            after the other without a fork record in between (pack_scene)
//   nrms   : 3 x float4 per triangle id {n0} {n1} {n2}   (read once per ray, for the closest hit only)
//   mats   : 3 x float4 per material {emission.xyz, type} {param0.xyz, alpha.x} {param1.xyz, alpha.y}
//   lights : 6 x float4 per light triangle {v0, materialId} {v1} {v2} {n0} {n1} {n2}
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "trav_asm.hip.h"
#include "scan_asm.hip.h"

namespace glrtx {

constexpr int kBlockThreads = 256;      // 4 wavefronts: a 16x16-pixel tile, one 8x8 sub-tile per wave
constexpr int kTile = 16;
constexpr int REF_ABSENT = INT32_MIN;
constexpr int REF_FIN = INT32_MIN;      // Trav::cur of a finished ray (nothing left on the stack)
constexpr int kMaxLdsMaterials = 256;   // 12 KiB of LDS at most
constexpr int kMaxLdsLights = 64;       // 6 KiB of LDS at most
constexpr int kTopForks = 128;          // forks of the top tree levels, numbered first (pack_scene): one contiguous 8 KiB block

constexpr float PT_EPS = 1.0e-4f;
constexpr float PT_INFTY = 1.0e8f;
constexpr float PT_PI = 3.14159274101257324f;
constexpr float PT_2PI = 6.28318548202514648f;  // the GLSL compiler folds 2.0*PI into one constant

struct DevScene {
    const float4 *forks;  // fork f at forks[4 f]; triangle t at forks[4 ~t] = forks[-4 (t + 1)]: ONE array with the triangle records
                          // stored (in reverse) in front of fork 0, so that a ref -- fork index or ~triangle -- is itself the
                          // signed record index and the traversal step needs no base-pointer select
    const float4 *nodes0; // first byte of that array (the all-zero record ~n_tri); record ref lies at byte ref * 64 + node_bias from
    unsigned node_bias;   // here: an unsigned 32-bit offset from a uniform base, which the load instructions take as they are
    const float4 *nrms;
    const float4 *mats;
    const float4 *lights;
    float4 root_lo, root_hi;  // the root fork's own box
    int root_ref;
    int root_boxed;     // the tree's root is a fork in the wire format: root_lo / root_hi are tested before the first step (also when the
                        // root is packed as a chained leaf pair and root_ref is a triangle record)
    int n_light;
    int n_mat;
    int n_fork;
    int stack_entries;  // per-lane traversal stack entries in LDS
    int mats_in_lds;    // 1: materials staged into LDS at kernel start
    int lights_in_lds;  // 1: the light triangles ({v0, material} {v1} {v2} {n0} {n1} {n2}, at most kMaxLdsLights) staged into LDS behind the materials (round 6): the six
                        //    gathers of every light sample (:344-352) become LDS reads
    int lds_head_f4;    // float4s staged at the head of a workgroup's LDS: the materials, then the lights
    // "Vine" trees -- every fork has a leaf as children.y: the brute-force scan of BASELINE config 3 expressed in the
    // node format (glrt_bvh_build_chain) -- are also stored as a list in visiting order and scanned, see trav_scan().
    const float4 *vine;  // records of 4 float4: {fork box min, v0.x} {fork box max, v0.y} {v0.z, v1-v0} {v2-v0, triangle}: the n_vine - 1 fork records, never-hit
                         // records up to index vine_main (a multiple of 4), the last leaf's record there, three never-hit records behind it (glrtx.hip: pack_scene)
    int n_vine;          // triangles in the list; 0: not a vine
    int vine_uniform;    // 1: all fork boxes are the same box (vine[0]'s)
    int vine_main;       // index of the last record
    int shadow_limited;  // 0: shadow rays are searched like the reference's (default); 1: with the range limit (shadow_limit(): opt-in, not exact)
};

struct KernelArgs {
    DevScene sc;
    float cam[34];           // camera block: c2w[16] (u_c2wMat), s2c[16] (u_s2cMat), u_apertureRadius, u_focalLength
    float seed_x, seed_y;
    int n_samples, max_depth;
    int width, height;       // full image
    int owned_rows;          // rows in this partition
    int rank, world, stripe; // row-stripe partition
    float4 *accum;           // owned_rows x pitch
    int pitch_f4;            // accumulator pitch in float4 units
    unsigned long long *ray_counter;
    int tiles_x, n_tiles;
    unsigned *hit_hist;      // null, or (glrtx_hit_histogram: a calibration frame) a counter per leaf record: closest hits of the path rays shaded in this launch
};

#define DEV __device__ __forceinline__

// Streamed data (path state, sample planes, the reads of ray records): written once and read once per trip, never reused from
// cache.  GLRTX_STREAM_NT marks those accesses non-temporal so that they do not displace the BVH from L2 (-2.3 % per frame).
// Two streams are deliberately NOT marked (per-stream A/B, profiles/r02_ab_flags.txt): the ray records' stores -- the same
// workgroup reads them back within a phase, and written plainly they are still cached then (-4 %) -- and the hit records
// (scattered 16-byte stores that a plain store lets merge in L2 now and then, -1.6 %).
#ifndef GLRTX_STREAM_NT
#define GLRTX_STREAM_NT 3
#endif
typedef float nfloat4 __attribute__((ext_vector_type(4)));
typedef float nfloat2 __attribute__((ext_vector_type(2)));
DEV float4 ld_stream(const float4 *p) {
#if GLRTX_STREAM_NT & 1
    const nfloat4 v = __builtin_nontemporal_load(reinterpret_cast<const nfloat4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
DEV float2 ld_stream(const float2 *p) {
#if GLRTX_STREAM_NT & 1
    const nfloat2 v = __builtin_nontemporal_load(reinterpret_cast<const nfloat2 *>(p));
    return make_float2(v.x, v.y);
#else
    return *p;
#endif
}
DEV void st_stream(float4 *p, float4 v) {
#if GLRTX_STREAM_NT & 2
    nfloat4 n; n.x = v.x; n.y = v.y; n.z = v.z; n.w = v.w;
    __builtin_nontemporal_store(n, reinterpret_cast<nfloat4 *>(p));
#else
    *p = v;
#endif
}
DEV void st_stream(float2 *p, float2 v) {
#if GLRTX_STREAM_NT & 2
    nfloat2 n; n.x = v.x; n.y = v.y;
    __builtin_nontemporal_store(n, reinterpret_cast<nfloat2 *>(p));
#else
    *p = v;
#endif
}

// ------------------------------------------------------------------------------------------ sin / cos
// Cephes sinf/cosf with FMA, the routine the reference's GL implementation uses for sin()/cos().
DEV float sincos_core(float xabs, int je, bool sin_poly) {
    const float yf = (float)je;
    float x = __builtin_fmaf(yf, -0.78515625f, xabs);
    x = __builtin_fmaf(yf, -2.4187564849853515625e-4f, x);
    x = __builtin_fmaf(yf, -3.77489497744594108e-8f, x);
    const float z = x * x;
    float s = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    s = __builtin_fmaf(s, z, -1.6666654611e-1f);
    s = s * z;
    s = __builtin_fmaf(s, x, x);
    float c = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    c = __builtin_fmaf(c, z, 4.166664568298827e-2f);
    c = c * z;
    c = c * z;
    c = c - z * 0.5f;
    c = c + 1.0f;
    return sin_poly ? s : c;
}
DEV float clamp_unit(float r, float x) {
    if ((__float_as_uint(x) & 0x7f800000u) == 0x7f800000u) return __uint_as_float(0x7fc00000u);
    r = r < 1.0f ? r : 1.0f;
    r = r > -1.0f ? r : -1.0f;
    return r;
}
DEV float pt_sin(float x) {
    const float xabs = __builtin_fabsf(x);
    const int j1 = (int)(xabs * 1.27323954473516f) + 1;
    const int je = j1 & ~1;
    const uint32_t sign = (__float_as_uint(x) ^ ((uint32_t)j1 << 29)) & 0x80000000u;
    const float r = sincos_core(xabs, je, (je & 2) == 0);
    return clamp_unit(__uint_as_float(__float_as_uint(r) ^ sign), x);
}
DEV float pt_cos(float x) {
    const float xabs = __builtin_fabsf(x);
    const int je = ((int)(xabs * 1.27323954473516f) + 1) & ~1;
    const int j2 = je - 2;
    const uint32_t sign = ((uint32_t)(~j2) & 4u) << 29;
    const float r = sincos_core(xabs, je, (j2 & 2) == 0);
    return clamp_unit(__uint_as_float(__float_as_uint(r) ^ sign), x);
}
// The RNG argument is bounded (|t| < 92: state, seed in [0,1)), so the inf/NaN guard and the
// [-1,1] clamp of the general routine cannot trigger differently; keep the clamp, drop nothing.

// ------------------------------------------------------------------------------------------ rand() :104-111
struct Rng {
    float x, y, sx, sy;
};
DEV float pt_rand(Rng &s) {
    const float a = 12.9898f, b = 78.233f, c = 43758.5453f;
    const float dy = (s.y - s.sy) * b;  // old state.y term, shared by both updates
    float t = dy + (s.x - s.sx) * a;
    float p = pt_sin(t) * c;
    s.x = p - __builtin_floorf(p);
    t = dy + (s.x - s.sx) * a;
    p = pt_sin(t) * c;
    s.y = p - __builtin_floorf(p);
    return s.x;
}

// ------------------------------------------------------------------------------------------ helpers
DEV float dot3(float ax, float ay, float az, float bx, float by, float bz) { return (az * bz + ay * by) + ax * bx; }
// Quotients.  The kernels run with fp32 denormals flushed on input and output (-fgpu-flush-denormals-to-zero): that is the reference's arithmetic -- llvmpipe's
// rasteriser threads set MXCSR FTZ | DAZ, and so does the oracle (pt_oracle.c: pt_render_rows).  In that mode the compiler's correctly rounded a / b switches
// the denormal mode on and off around its core (two s_setreg per quotient), and three short forms are PROVABLY the same value -- tools/ubench/rcp_exact.hip runs
// every one of the 2^32 float bit patterns through each of them on the device against the compiler's quotient (tests/test_gpu_parity.py runs it): 0 mismatches.
//   rcp_newton(x): v_rcp_f32 and ONE Newton step with fused multiply-adds = 1.0f / x for every normal finite x (beyond 2^126 both give the flushed zero of x's
//                  sign).  Zeros, denormals and infinities come out as NaN instead of inf / 0: the triangle tests, its only users, reject |det| < EPS before
//                  anything reads the quotient, and an infinite det changes nothing either way (t comes out as 0 or NaN: never a hit that is closer).
//   frcp(x):       the same, with the raw v_rcp_f32 result where x is a zero, a denormal or an infinity: = 1.0f / x for EVERY bit pattern (a NaN stays a NaN).
//   div_pi(x):     x * RN(1 / PI) corrected by one residual step = x / PI for every x that is +0 or has 2^-100 <= |x| <= 2^120; a wave that holds any other
//                  value takes the full division (a branch, not a select between both).
DEV float rcp_newton(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
DEV float frcp(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    const float n = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
    return __builtin_amdgcn_class(x, 0x2F4) ? r : n;  // -inf, -denormal, -0, +0, +denormal, +inf
}
constexpr float PT_INV_PI = 0.318309873342514038f;  // RN(1 / PT_PI)
DEV float div_pi(float x) {
    const unsigned a = __float_as_uint(x) & 0x7FFFFFFFu;
    // (-0 counts as out of range since round 5: the residual step turns it into +0 -- fma(-PI, -0, -0) = +0 -- found when tools/ubench/rcp_exact.hip began to check the
    //  short form on every pattern of its range by itself; the wave-level branch had hidden it, the checker's waves being runs of consecutive patterns)
    if (__any(__float_as_uint(x) != 0u && (a - 0x0D800000u) > (0x7B800000u - 0x0D800000u))) return x / PT_PI;  // -0, 0 < |x| < 2^-100, |x| > 2^120, inf, NaN
    const float q = x * PT_INV_PI;
    return __builtin_fmaf(__builtin_fmaf(-PT_PI, q, x), PT_INV_PI, q);
}
DEV float fdiv(float a, float b) { return a / b; }
DEV float rsq(float x) { return frcp(__builtin_sqrtf(x)); }  // IEEE sqrt then IEEE reciprocal
// GLSL min/max as the reference's GL implementation lowers them (other operand on NaN):
DEV float fmin_g(float a, float b) { return (b != b) ? a : (a < b ? a : b); }
DEV float fmax_g(float a, float b) { return (b != b) ? a : (a > b ? a : b); }
DEV float fmin_c(float x, float c) { return x < c ? x : c; }  // one operand constant
DEV float fmax_c(float x, float c) { return x > c ? x : c; }

// Explicit address spaces: left generic, the two arms are merged into one flat_load through a selected pointer, and a flat
// access to LDS takes the vector-memory path instead of ds_read.
typedef const __attribute__((address_space(3))) nfloat4 *lds_cf4;
typedef const __attribute__((address_space(1))) nfloat4 *glb_cf4;
typedef const __attribute__((address_space(1))) nfloat2 *glb_cf2;
DEV float4 to_f4(nfloat4 v) { return make_float4(v.x, v.y, v.z, v.w); }
struct Hit {
    float t;    // INFTY on a miss
    int tri;    // closest triangle, -1 on a miss
    float u, v; // barycentrics of the closest hit
};

// ------------------------------------------------------------------------------------------ intersect :276-335
// Iterative DFS in the reference's order (push children.x, push children.y, pop y first,
// :299-307): "continue with y, stack x".  The per-lane stack lives in LDS, entry e of lane l at
// stack[e * kBlockThreads + l] (bank = l mod 32: conflict-free ds_read/write_b32).
// Box test of one child, intersectBBox :259-274 + the cull of :298.  v_min/v_max return the non-NaN
// operand, which is the NaN rule needed here; the sign of a zero cannot change the comparison.
DEV bool box_pass(float4 lo, float4 hi, float ox, float oy, float oz, float ix, float iy, float iz, float tHit, float &t0) {
    const float fx = (hi.x - ox) * ix, fy = (hi.y - oy) * iy, fz = (hi.z - oz) * iz;
    const float nx = (lo.x - ox) * ix, ny = (lo.y - oy) * iy, nz = (lo.z - oz) * iz;
    const float t1 = __builtin_fminf(__builtin_fmaxf(fx, nx), __builtin_fminf(__builtin_fmaxf(fy, ny), __builtin_fmaxf(fz, nz)));
    t0 = __builtin_fmaxf(__builtin_fminf(fx, nx), __builtin_fmaxf(__builtin_fminf(fy, ny), __builtin_fminf(fz, nz)));
    return __builtin_fminf(t1, tHit) >= t0;  // (t1 >= t0 && t0 <= tHit) is evaluated as min(t1, tHit) >= t0
}

// Iterative DFS in the reference's order (push children.x, push children.y, pop y first, :299-307):
// "continue with y, stack x".  A fork node carries the boxes of its two children, so one 64-byte
// fetch decides both (the reference fetches a child, then tests the child's own box: the same two
// tests, one dependent memory round trip earlier).  Testing the stacked child early uses a tHit that
// may still shrink; the entry keeps its entry distance t0 and is re-checked against the current tHit
// when popped, which reproduces the reference's visit set exactly (its test at pop time is
// t1 >= t0 -- independent of tHit -- and t0 <= tHit).  Leaf children are never box-tested, as in the
// reference (:310-331).  The per-lane stack lives in LDS as 8-byte {t0, ref} entries, entry e of lane l
// at byte (e * kBlockThreads + l) * 8: one conflict-free ds_read/write_b64 per pop/push.
#ifdef GLRTX_TRAV_STATS
// Diagnostic build only (-DGLRTX_TRAV_STATS): [0] wave loop iterations, [1] active lanes summed over
// iterations, [2] lanes on the fork path, [3] lanes on the leaf path, [4] iterations with both paths live
__device__ unsigned long long g_trav_stats[8];
__device__ unsigned long long g_trav_hist[16];  // rays by ceil(log2(iterations))
__device__ unsigned long long g_trav_sp_hist[16];  // lane-steps by the stack pointer at the step's start (15: 15 or more) -- sizes a short LDS stack
__device__ unsigned long long g_trav_trips[4];  // [0] stepping trips, [1] lanes with a ray at their start, [2] trips after the queue ran out (drain), [3] lanes in those
// [5] distinct 64-byte node records, [6] distinct 128-byte lines fetched by the wave (summed over iterations), [7] lanes carrying a path ray
DEV void trav_stats_iter(int cur, const void *rec, bool path_ray, int sp) {
    for (int d = 0; d < 16; d++) {
        const unsigned long long md = __ballot(d < 15 ? sp == d : sp >= 15);
        if (md != 0ull && (int)(threadIdx.x & 63) == __ffsll((long long)md) - 1) atomicAdd(&g_trav_sp_hist[d], (unsigned long long)__popcll(md));
    }
    const unsigned long long m = __ballot(1), mf = __ballot(cur >= 0), mp = __ballot(path_ray);
    int n_rec = 0, n_line = 0;
    const unsigned lo = (unsigned)((uintptr_t)rec >> 6), hi = (unsigned)((uintptr_t)rec >> 38);
    for (unsigned long long rem = m; rem != 0ull;) {
        const int l = __ffsll((long long)rem) - 1;
        const unsigned vlo = (unsigned)__shfl((int)lo, l), vhi = (unsigned)__shfl((int)hi, l);
        rem &= ~__ballot(lo == vlo && hi == vhi);
        n_rec++;
    }
    for (unsigned long long rem = m; rem != 0ull;) {
        const int l = __ffsll((long long)rem) - 1;
        const unsigned vlo = (unsigned)__shfl((int)(lo >> 1), l), vhi = (unsigned)__shfl((int)hi, l);
        rem &= ~__ballot((lo >> 1) == vlo && hi == vhi);
        n_line++;
    }
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
        atomicAdd(&g_trav_stats[0], 1ull);
        atomicAdd(&g_trav_stats[1], (unsigned long long)__popcll(m));
        atomicAdd(&g_trav_stats[2], (unsigned long long)__popcll(mf));
        atomicAdd(&g_trav_stats[3], (unsigned long long)__popcll(m & ~mf));
        if (mf != 0 && (m & ~mf) != 0) atomicAdd(&g_trav_stats[4], 1ull);
        atomicAdd(&g_trav_stats[5], (unsigned long long)n_rec);
        atomicAdd(&g_trav_stats[6], (unsigned long long)n_line);
        atomicAdd(&g_trav_stats[7], (unsigned long long)__popcll(mp));
    }
}
#endif

// Traversal state of one ray: the DFS above as an explicit state machine, so that the megakernels
// (run it to completion) and the wavefront traversal kernel (one step per loop trip, lanes refilled
// with fresh rays as they finish) execute literally the same code.
// Shadow rays (sampleDirect, :337-403) are closest-hit queries in the reference, but their result is used for one thing
// only: "hit && |dist - tHit| < EPS" (:367).  That outcome is settled early, EXACTLY, in one way:
//  * once any hit with dist - t >= EPS is known the test has failed: the final tHit can only be smaller, and rounding is
//    monotonic -- so the traversal stops there.  The search is otherwise the reference's own: tHit starts at INFTY and the
//    boxes are culled by the hits found, in the reference's visiting order.
// OPT-IN since round 5 (glrtx_set_shadow_range_limit(ctx, 1) / GLRTX_SHADOW_LIMIT=1; it was the default in rounds 1-4), NOT exact:
//  * range limit: no hit at or beyond dist + EPS can be accepted, and if the closest hit lies out there the test fails whatever
//    it is -- so the search starts with tHit = `limit` instead of INFTY and reports a miss in that case.  The limit also culls
//    BOXES, and the outcome is the reference's only as long as no triangle's COMPUTED t lies below the limit while the computed
//    entry distance of one of its ancestors' boxes lies above it.  A box's entry is good to ~3 ulps, but a Moeller-Trumbore t is
//    a quotient of two cancelling triple products: its relative error grows like 2^-23 / cos(angle to the triangle's plane), and the
//    test only rejects |det| < EPS -- for a grazing shadow ray that ends near the edge of a light lying flush in its box no margin
//    in terms of dist (2 EPS in rounds 1-3, + 2^-13 dist in round 4: profiles/r04_ab_shadow_limit.txt) is a bound.  Rare (no
//    fixture, fuzz case or full-size frame has shown it since the relative margin), but not provably absent: hence opt-in.
//    Cost of the exact search: profiles/r05_ab_shadow_limit.txt.
// A path's own rays use limit = INFTY and stop_d = -inf, which switches both off.
#ifndef GLRTX_SHADOW_REL
#define GLRTX_SHADOW_REL 0x1p-13f
#endif
// `limited` = DevScene::shadow_limited: 0 (default) -- no range limit, the result is INFTY; 1 -- the limit above.
DEV float shadow_limit(float dist, int limited) {
    float m = __builtin_fmaxf(dist + 2.0f * PT_EPS, __uint_as_float(__float_as_uint(dist) + 1u));
    m = __builtin_fmaxf(m, __builtin_fmaf(dist, GLRTX_SHADOW_REL, dist));
    return limited ? fmin_c(m, PT_INFTY) : PT_INFTY;  // NaN distance: INFTY, i.e. the unrestricted search
}

struct Trav {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz;
    Hit h;
    float stop_d;  // shadow rays: the light sample's distance; others: -inf
    int cur, sp;
#ifdef GLRTX_TRAV_STATS
    unsigned iters;
#endif
};

// Returns false if the ray is finished before the first step (root box missed).
// `root` = {root_lo, root_hi}: &sc.root_lo, or the workgroup's LDS copy of it (the wavefront kernel keeps launch constants
// that only the refill and camera code read out of the scalar registers, which its traversal loop needs for itself).
DEV bool trav_init(const DevScene &sc, const float4 *root, Trav &T, float ox, float oy, float oz, float dx, float dy, float dz,
                   float limit = PT_INFTY, float stop_d = -__builtin_inff()) {
    T.ox = ox; T.oy = oy; T.oz = oz; T.dx = dx; T.dy = dy; T.dz = dz;
    T.ix = frcp(dx); T.iy = frcp(dy); T.iz = frcp(dz);  // :260 (loop-invariant there)
    T.h.t = limit; T.h.tri = -1; T.h.u = 0.f; T.h.v = 0.f;
    T.stop_d = stop_d;
    T.sp = 0;
    T.cur = sc.root_ref;
#ifdef GLRTX_TRAV_STATS
    T.iters = 0;
#endif
    if (sc.root_boxed) {  // the root fork's own box
        float t0;
        if (!box_pass(root[0], root[1], ox, oy, oz, T.ix, T.iy, T.iz, T.h.t, t0)) return false;
    }
    return true;
}

// One trip of the traversal loop: process T.cur (fork or leaf), then pick the next node.
// Returns true when the ray is finished (stack empty).
// (Top tree levels staged in LDS were built and measured in round 2: worth nothing, profiles/r02_lds_top.json -- the cost of a wave's node fetch is set by its number of
// distinct cache lines (profiles/r02_ubench_gather.json), and the lanes at the top levels share theirs with many others; removed.)
// What that experiment did find: written as below, every lane fetches its whole 56-byte record with FOUR load instructions
// (dwordx4, dwordx4, dwordx3, dwordx3) issued together; the previous form -- three loads for all lanes, then one more and the
// two refs as single dwords on the fork arm, six instructions -- was 8 % slower per frame for fewer bytes.
template <bool CLOSEST>
DEV bool trav_step(const DevScene &sc, int *stack, Trav &T) {
#ifdef GLRTX_TRAV_STATS
    trav_stats_iter(T.cur, (const void *)(sc.forks + 4 * (ptrdiff_t)T.cur), T.stop_d == -__builtin_inff(), T.sp);
    T.iters++;
#define TS_DONE atomicAdd(&g_trav_hist[T.iters <= 1 ? 0 : (32 - __clz((int)T.iters - 1)) > 15 ? 15 : (32 - __clz((int)T.iters - 1))], 1ull)
#else
#define TS_DONE
#endif
    // Both arms are written with straight-line arithmetic and ONE combined predicate instead of the
    // reference's nested early-outs: in a divergent wave some lane takes every early-out path anyway, and
    // each nesting level costs scalar exec-mask bookkeeping on the latency-critical instruction stream.
    // Evaluating a test's later terms when an earlier one already failed cannot change the outcome
    // (pure IEEE arithmetic; a NaN/inf produced behind a failed test is masked by the predicate).
    const int cur = T.cur;
    bool need_pop = true;
    // Fork and triangle records have the same 64-byte shape and are fetched by the SAME four loads, issued
    // before the wave splits into its fork lanes and its triangle lanes: in a mixed wave (3 of 4 iterations)
    // the two arms then cost one memory round trip, not two.
    const bool is_fork = cur >= 0;
    const float4 *N = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sc.nodes0) + ((unsigned)cur * 64u + sc.node_bias));
    // Every lane fetches its whole record here -- dwordx4, dwordx4, dwordx3, dwordx3: FOUR load instructions issued together (the
    // compiler drops the unused C.w / D.w).  The vector-memory pipe charges per instruction and per distinct cache line, not per
    // byte: fetching the fork arm's second box and refs separately (six instructions, 20 bytes less for a triangle lane) was 8 %
    // slower per frame (profiles/r02_lds_top.json, r02_ubench_gather.json).
    float4 A = N[0], B = N[1], C = N[2], D = N[3];
    // Pin the fork-only words (the two refs, the second child's far corner) here, in front of the arms: without it the compiler
    // sinks their loads into the fork arm and the fetch becomes six instructions instead of four (8 % slower per frame).
    asm volatile("" : "+v"(A.w), "+v"(B.w), "+v"(D.x), "+v"(D.y), "+v"(D.z));
    // Two separate ifs, fork arm first: its extra loads (second box, refs) must go out BEFORE the triangle arithmetic of the
    // wave's leaf lanes.  (As one if / else the compiler may place the leaf arm first -- it did once the address select was
    // gone -- and the fork lanes then wait a second round trip behind it: 5 % of the frame.)
    if (is_fork) {
        const int l = __float_as_int(A.w), r = __float_as_int(B.w);
        float t0l, t0r;
        const bool bl = box_pass(A, B, T.ox, T.oy, T.oz, T.ix, T.iy, T.iz, T.h.t, t0l);
        const bool br = box_pass(C, D, T.ox, T.oy, T.oz, T.ix, T.iy, T.iz, T.h.t, t0r);
        // leaf children are never box-tested (:310-331): in the packed record a leaf child -- and an absent one, the never-hit
        // record ~n_tri -- carries the box (-inf, +inf), which passes by itself with t0 = -inf (pack_scene)
        const bool pl = bl;
        const bool pr = br;
        if (pl && pr) {  // continue with the right child, the left one waits on the stack
            reinterpret_cast<int2 *>(stack)[T.sp * kBlockThreads] = make_int2(__float_as_int(t0l), l);  // {t0, ref}: one ds_write_b64
            T.sp++;
        }
        T.cur = pr ? r : l;
        need_pop = !(pl || pr);
    }
    if (!is_fork) {
        // leaf :310-331 with intersect(Ray, Triangle) :226-257; A = {v0, material}, B = v1-v0, C = v2-v0
        const int t = ~cur;
        const float tx = T.ox - A.x, ty = T.oy - A.y, tz = T.oz - A.z;
        const float px = T.dy * C.z - T.dz * C.y;
        const float py = T.dz * C.x - T.dx * C.z;
        const float pz = T.dx * C.y - T.dy * C.x;
        const float det = dot3(B.x, B.y, B.z, px, py, pz);
        const float U = dot3(tx, ty, tz, px, py, pz);
        const float inv = rcp_newton(det);
        const float u = U * inv;
        const float qx = ty * B.z - tz * B.y;
        const float qy = tz * B.x - tx * B.z;
        const float qz = tx * B.y - ty * B.x;
        const float V = dot3(T.dx, T.dy, T.dz, qx, qy, qz);
        const float v = V * inv;
        const float tt = dot3(C.x, C.y, C.z, qx, qy, qz) * inv;
        // (&& / || on purpose: the compiler turns the later terms into a branch that a wave skips when none of its leaf lanes
        //  is still in the running -- most tested triangles are missed; with & and | the step was 7 % slower)
        const bool hit = !(-PT_EPS < det && det < PT_EPS) && !(u < 0.0f || 1.0f < u) &&
                         !(v < 0.0f || 1.0f < inv * (U + V)) &&  // u+v>1 is evaluated as inv*(U+V)>1
                         !(PT_EPS >= tt);
        const bool closer = hit && tt < T.h.t;  // strict: among equal distances the first one visited wins (:325)
        T.h.tri = closer ? t : T.h.tri;
        if (CLOSEST) { T.h.u = closer ? u : T.h.u; T.h.v = closer ? v : T.h.v; }
        T.h.t = closer ? tt : T.h.t;  // == hit ? min(tHit, tt) : tHit (a NaN tt is never closer)
        // shadow ray: once an occluder is known the light test has failed and the traversal ends; otherwise on to the triangle chained
        // behind this one (the other leaf of a leaf pair, pack_scene) or, without one, to the stack
        const bool stopped = T.stop_d - T.h.t >= PT_EPS;
        const int next = __float_as_int(B.w);
        T.cur = stopped ? REF_FIN : next;
        need_pop = !stopped && next == REF_FIN;
    }
    // pop; entries whose entry distance now lies beyond tHit are the ones the reference culls at :298.  Hand-written: as C++ the
    // compiler's structurizer spends ~30 instructions of exec-mask bookkeeping per trip on this loop-with-two-exits, the loop
    // itself is 10.  Lanes drop out of the loop when their stack is empty -- their ref becomes REF_FIN: the ray is finished -- or
    // when the entry they read survives.
    if (need_pop) {
        float t0;
        int sp = T.sp, ref = REF_FIN;
        unsigned long long save;
        unsigned addr;
        const unsigned lds_base = (unsigned)(uintptr_t)stack;
        asm volatile(
            "s_mov_b64 %[save], exec\n"
            "1:\n\t"
            "v_cmp_ne_u32 vcc, 0, %[sp]\n\t"
            "v_cndmask_b32 %[ref], %[fin], %[ref], vcc\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "s_cbranch_execz 2f\n\t"
            "v_add_u32 %[sp], -1, %[sp]\n\t"
            "v_lshl_add_u32 %[addr], %[sp], 11, %[base]\n\t"
            "ds_read_b32 %[ref], %[addr] offset:4\n\t"
            "ds_read_b32 %[t0], %[addr]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_cmp_gt_f32 vcc, %[t0], %[th]\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "s_cbranch_execnz 1b\n"
            "2:\n\t"
            "s_mov_b64 exec, %[save]"
            : [t0] "=&v"(t0), [ref] "+&v"(ref), [sp] "+&v"(sp), [save] "=&s"(save), [addr] "=&v"(addr)  // (early-clobber: ref starts as
            : [base] "v"(lds_base), [th] "v"(T.h.t), [fin] "v"(REF_FIN)                                   //  REF_FIN and must not share fin's register)
            : "vcc", "scc", "memory");  // (exec is restored; s_and_b64 writes scc)
        static_assert(kBlockThreads * 8 == 1 << 11, "the pop loop shifts the stack index by 11: entry e of lane l at byte (e * kBlockThreads + l) * 8");
        T.sp = sp;
        T.cur = ref;
    }
#ifdef GLRTX_TRAV_STATS
    if (T.cur == REF_FIN) { TS_DONE; }
#endif
    return T.cur == REF_FIN;
}

// Up to GLRTX_STEPS_PER_TRIP steps of trav_step<true> for the lanes enabled on entry, as one hand-written asm statement
// (trav_asm.hip.h); a lane leaves early when its ray is finished (T.cur == REF_FIN).  Used by the wavefront kernel.
#ifdef GLRTX_STEP_TIMING
// Diagnostic build only: [0] lane-steps, [1] shader clocks of those steps summed per lane, [2] the part spent waiting for the node fetch
__device__ unsigned long long g_step_timing[4];
struct StepTiming { unsigned t = 0, w = 0, n = 0; };
#define GLRTX_TS_PARAM , StepTiming &ST
#define GLRTX_TS_OPERANDS , [tt] "+&v"(ST.t), [tw] "+&v"(ST.w), [tn] "+&v"(ST.n)
#define GLRTX_TS_CLOBBERS , "s90", "s91", "s92", "s93", "s94", "s95"
#else
#define GLRTX_TS_PARAM
#define GLRTX_TS_OPERANDS
#define GLRTX_TS_CLOBBERS
#endif
// FETCH: 0 one record per lane, 1 pair-cooperative, 2 the two forms in alternate steps (pair first) -- see trav_asm.hip.h and launch_wgwf (glrtx.hip)
template <int FETCH>
DEV void trav_steps_asm(const DevScene &sc, int *stack, Trav &T GLRTX_TS_PARAM) {
    unsigned long long s_entry, s_act, s_leaf, s_bl, s_br, s_pop, s_tmp;
    const unsigned stk = (unsigned)(uintptr_t)stack;
    static_assert(kBlockThreads * 8 == 1 << 11, "trav_asm.hip.h shifts the stack index by 11: entry e of lane l at byte (e * kBlockThreads + l) * 8");
    static_assert(REF_FIN == INT32_MIN, "trav_asm.hip.h materialises REF_FIN with v_bfrev_b32 v, 1");
    // pair-cooperative fetch: which 16-byte pieces a lane reads of the even lane's record (0 and 2, odd lanes 1 and 3) and of the odd lane's (1 and 3, odd lanes 0 and 2)
    const unsigned par16 = (threadIdx.x & 1u) << 4;
    const unsigned bias_e = sc.node_bias + par16, bias_o = sc.node_bias + 16u - par16;
    if (FETCH == 2) {  // pair, lane, pair, lane, ...: the pipe and the SIMDs take turns at being the busier unit (-1.3 % on the 10 k-triangle tree)
        asm volatile(
            "s_mov_b64 %[entry], exec\n\t"
            "s_mov_b64 %[act], exec\n\t"
            GLRTX_ASM_SET_VBASE
            "v_bfrev_b32 v[GLRTX_VB+11], 1\n\t"
            GLRTX_REP(GLRTX_STEPS_PER_TRIP_HALF, GLRTX_TRAV_STEP_ASM_PAIR GLRTX_TRAV_STEP_ASM_LANE)
            "99:\n\t"
            "s_mov_b64 exec, %[entry]"
            : [th] "+&v"(T.h.t), [tri] "+&v"(T.h.tri), [hu] "+&v"(T.h.u), [hv] "+&v"(T.h.v), [cur] "+&v"(T.cur), [sp] "+&v"(T.sp),
              [entry] "=&s"(s_entry), [act] "=&s"(s_act), [leaf] "=&s"(s_leaf), [bl] "=&s"(s_bl), [br] "=&s"(s_br), [pop] "=&s"(s_pop), [tmp] "=&s"(s_tmp) GLRTX_TS_OPERANDS
            : [ox] "v"(T.ox), [oy] "v"(T.oy), [oz] "v"(T.oz), [dx] "v"(T.dx), [dy] "v"(T.dy), [dz] "v"(T.dz), [ix] "v"(T.ix), [iy] "v"(T.iy), [iz] "v"(T.iz),
              [sd] "v"(T.stop_d), [stk] "v"(stk), [base] "s"(sc.nodes0), [bias] "s"(sc.node_bias), [eps] "s"(PT_EPS),
              [odd] "s"(0xAAAAAAAAAAAAAAAAull), [biase] "v"(bias_e), [biaso] "v"(bias_o)
            : "vcc", "scc", "memory", GLRTX_ASM_VCLOBBERS_PAIR GLRTX_TS_CLOBBERS);
        return;
    }
    if (FETCH == 1) {
        asm volatile(
            "s_mov_b64 %[entry], exec\n\t"
            "s_mov_b64 %[act], exec\n\t"
            GLRTX_ASM_SET_VBASE
            "v_bfrev_b32 v[GLRTX_VB+11], 1\n\t"
            GLRTX_REP(GLRTX_STEPS_PER_TRIP, GLRTX_TRAV_STEP_ASM_PAIR)
            "99:\n\t"
            "s_mov_b64 exec, %[entry]"
            : [th] "+&v"(T.h.t), [tri] "+&v"(T.h.tri), [hu] "+&v"(T.h.u), [hv] "+&v"(T.h.v), [cur] "+&v"(T.cur), [sp] "+&v"(T.sp),
              [entry] "=&s"(s_entry), [act] "=&s"(s_act), [leaf] "=&s"(s_leaf), [bl] "=&s"(s_bl), [br] "=&s"(s_br), [pop] "=&s"(s_pop), [tmp] "=&s"(s_tmp) GLRTX_TS_OPERANDS
            : [ox] "v"(T.ox), [oy] "v"(T.oy), [oz] "v"(T.oz), [dx] "v"(T.dx), [dy] "v"(T.dy), [dz] "v"(T.dz), [ix] "v"(T.ix), [iy] "v"(T.iy), [iz] "v"(T.iz),
              [sd] "v"(T.stop_d), [stk] "v"(stk), [base] "s"(sc.nodes0), [eps] "s"(PT_EPS),
              [odd] "s"(0xAAAAAAAAAAAAAAAAull), [biase] "v"(bias_e), [biaso] "v"(bias_o)
            : "vcc", "scc", "memory", GLRTX_ASM_VCLOBBERS_PAIR GLRTX_TS_CLOBBERS);
        return;
    }
    asm volatile(
        "s_mov_b64 %[entry], exec\n\t"
        "s_mov_b64 %[act], exec\n\t"
        GLRTX_ASM_SET_VBASE
        "v_bfrev_b32 v[GLRTX_VB+11], 1\n\t"
        GLRTX_REP(GLRTX_STEPS_PER_TRIP, GLRTX_TRAV_STEP_ASM_LANE)
        "99:\n\t"
        "s_mov_b64 exec, %[entry]"
        : [th] "+&v"(T.h.t), [tri] "+&v"(T.h.tri), [hu] "+&v"(T.h.u), [hv] "+&v"(T.h.v), [cur] "+&v"(T.cur), [sp] "+&v"(T.sp),
          [entry] "=&s"(s_entry), [act] "=&s"(s_act), [leaf] "=&s"(s_leaf), [bl] "=&s"(s_bl), [br] "=&s"(s_br), [pop] "=&s"(s_pop), [tmp] "=&s"(s_tmp) GLRTX_TS_OPERANDS
        : [ox] "v"(T.ox), [oy] "v"(T.oy), [oz] "v"(T.oz), [dx] "v"(T.dx), [dy] "v"(T.dy), [dz] "v"(T.dz), [ix] "v"(T.ix), [iy] "v"(T.iy), [iz] "v"(T.iz),
          [sd] "v"(T.stop_d), [stk] "v"(stk), [base] "s"(sc.nodes0), [bias] "s"(sc.node_bias), [eps] "s"(PT_EPS)
        : "vcc", "scc", "memory", GLRTX_ASM_VCLOBBERS GLRTX_TS_CLOBBERS);
}

// intersect(Ray, Triangle) :226-257 against the running closest hit; v0 / e1 = v1-v0 / e2 = v2-v0.  Used by the list scan only (the tree
// kernels carry the test in their step).  All lanes of a wave test the SAME triangle here, and most triangles are missed by all of
// them: the test leaves as soon as no lane of the wave can still hit -- after u (22 of the ~60 vector instructions), after v -- which
// skips operations whose results nothing would have read; the ones that are executed are the same, in the same order.
template <bool CLOSEST>
DEV void tri_test(Hit &h, int t, float ox, float oy, float oz, float dx, float dy, float dz, float v0x, float v0y, float v0z, float e1x,
                  float e1y, float e1z, float e2x, float e2y, float e2z) {
    const float tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
    const float px = dy * e2z - dz * e2y;
    const float py = dz * e2x - dx * e2z;
    const float pz = dx * e2y - dy * e2x;
    const float det = dot3(e1x, e1y, e1z, px, py, pz);
    const float U = dot3(tx, ty, tz, px, py, pz);
    const float inv = rcp_newton(det);  // (|det| < EPS is rejected below whatever its reciprocal is)
    const float u = U * inv;
    const bool ok_u = !(-PT_EPS < det && det < PT_EPS) && !(u < 0.0f || 1.0f < u);
    if (!__any(ok_u)) return;
    const float qx = ty * e1z - tz * e1y;
    const float qy = tz * e1x - tx * e1z;
    const float qz = tx * e1y - ty * e1x;
    const float V = dot3(dx, dy, dz, qx, qy, qz);
    const float v = V * inv;
    const bool ok_v = ok_u && !(v < 0.0f || 1.0f < inv * (U + V));
    if (!__any(ok_v)) return;
    const float tt = dot3(e2x, e2y, e2z, qx, qy, qz) * inv;
    const bool hit = ok_v && !(PT_EPS >= tt);
    const bool closer = hit && tt < h.t;
    h.tri = closer ? t : h.tri;
    if (CLOSEST) { h.u = closer ? u : h.u; h.v = closer ? v : h.v; }
    h.t = hit ? __builtin_fminf(h.t, tt) : h.t;
}

// Traversal of a vine = a scan of its list in the order the reference's DFS meets the nodes: fork i (its own box is
// tested against the current tHit, :296-298; a failed test pushes nothing, which ends the traversal), then the
// triangle hanging off it; the last record is the final fork's other leaf, reached without a test (infinite box).
// All lanes of a wave walk the same list position: the record address is wave-uniform, so records come in through the scalar
// cache and are broadcast -- no vector-memory traffic, no stack.  List layout: DevScene::vine.
// A list whose forks all have the same box (what glrt_bvh_build_chain emits: BASELINE config 3) is scanned by the hand-written
// loop of scan_asm.hip.h; this C++ statement serves every other vine, and as the form the assembly is checked against (-DGLRTX_SCAN_CXX).
// Precondition: stop_d - limit < EPS (see scan_asm.hip.h).
template <bool CLOSEST>
DEV Hit trav_scan(const DevScene &sc, float ox, float oy, float oz, float dx, float dy, float dz, bool valid,
              float limit = PT_INFTY, float stop_d = -__builtin_inff()) {
    Hit h;
    h.t = limit; h.tri = -1; h.u = 0.f; h.v = 0.f;
    const float ix = frcp(dx), iy = frcp(dy), iz = frcp(dz);
    bool alive = valid;
    float t0u = 0.f, t1u = 0.f;
    if (sc.vine_uniform) {  // one box for every fork: its slab interval is a per-ray constant
        const float4 lo = sc.vine[0], hi = sc.vine[1];
        const float fx = (hi.x - ox) * ix, fy = (hi.y - oy) * iy, fz = (hi.z - oz) * iz;
        const float nx = (lo.x - ox) * ix, ny = (lo.y - oy) * iy, nz = (lo.z - oz) * iz;
        t1u = __builtin_fminf(__builtin_fmaxf(fx, nx), __builtin_fminf(__builtin_fmaxf(fy, ny), __builtin_fmaxf(fz, nz)));
        t0u = __builtin_fmaxf(__builtin_fminf(fx, nx), __builtin_fmaxf(__builtin_fminf(fy, ny), __builtin_fminf(fz, nz)));
#ifndef GLRTX_SCAN_CXX
        if (valid) {
            unsigned groups = (unsigned)sc.vine_main >> 2;
            unsigned long long s_entry, s_alive, s_tmp;
            asm volatile(GLRTX_SCAN_UNIFORM_ASM
                         : [th] "+&v"(h.t), [tri] "+&v"(h.tri), [hu] "+&v"(h.u), [hv] "+&v"(h.v), [grp] "+&s"(groups), [entry] "=&s"(s_entry), [alive] "=&s"(s_alive),
                           [tmp] "=&s"(s_tmp)
                         : [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [dx] "v"(dx), [dy] "v"(dy), [dz] "v"(dz), [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), [sd] "v"(stop_d),
                           [t0u] "v"(t0u), [t1u] "v"(t1u), [ptr] "s"(sc.vine), [eps] "s"(PT_EPS)
                         : "vcc", "scc", "memory", GLRTX_ASM_VCLOBBERS, GLRTX_SCAN_SCLOBBERS);
        }
        return h;
#endif
    }
    const int n = sc.n_vine, last_at = sc.vine_main;
    // Two records in flight while two are worked on (four sets of 16 SGPRs; as plain loads the compiler would fetch the records on the vector-memory path, per
    // lane).  The waits are explicit because the compiler does not count loads issued from inline asm; scalar loads return out of order, so the only wait there
    // is is lgkmcnt(0), and a load is covered by the work issued between it and that wait.  i: position in the visiting order; the last leaf's record lies at
    // index vine_main, never-hit records around it (positions past the end read those; they are not stepped).
    typedef float rec_t __attribute__((ext_vector_type(16)));
    auto issue = [&](rec_t &r, int i) { asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(r) : "s"(sc.vine + 4 * (size_t)(i < n - 1 ? i : last_at + (i - (n - 1)))) : "memory"); };
    auto pair_arrive = [&](rec_t &a, rec_t &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b) : : "memory"); };
    auto step = [&](const rec_t &r, int i) {
        if (alive) {
            bool pass;
            if (sc.vine_uniform && i + 1 < n) pass = __builtin_fminf(t1u, h.t) >= t0u;
            else {
                float t0;
                pass = box_pass(make_float4(r[0], r[1], r[2], 0.f), make_float4(r[4], r[5], r[6], 0.f), ox, oy, oz, ix, iy, iz, h.t, t0);
            }
            alive = pass;
            if (pass) {
                tri_test<CLOSEST>(h, __float_as_int(r[15]), ox, oy, oz, dx, dy, dz, r[3], r[7], r[8], r[9], r[10], r[11], r[12], r[13], r[14]);
                if (stop_d - h.t >= PT_EPS) alive = false;  // shadow ray: occluded for certain (see Trav)
            }
        }
    };
    rec_t a0, a1, b0, b1;
    issue(a0, 0);
    issue(a1, 1);
    pair_arrive(a0, a1);
    for (int i = 0; i < n; i += 4) {
        if (!__any(alive)) break;
        issue(b0, i + 2);
        issue(b1, i + 3);
        step(a0, i);
        if (i + 1 < n) step(a1, i + 1);
        pair_arrive(b0, b1);
        if (i + 2 >= n) break;
        issue(a0, i + 4);
        issue(a1, i + 5);
        step(b0, i + 2);
        if (i + 3 < n) step(b1, i + 3);
        pair_arrive(a0, a1);
    }
    return h;
}

template <bool CLOSEST>
DEV Hit traverse(const DevScene &sc, int *stack, float ox, float oy, float oz, float dx, float dy, float dz,
                 float limit = PT_INFTY, float stop_d = -__builtin_inff()) {
    Trav T;
    if (trav_init(sc, &sc.root_lo, T, ox, oy, oz, dx, dy, dz, limit, stop_d))
        while (!trav_step<CLOSEST>(sc, stack, T)) {}
    return T.h;
}

// fresnelConductor :158-178, one channel
DEV float fresnel1(float c2, float s2, float cosI, float eta, float k) {
    const float eta2 = eta * eta, k2 = k * k;
    const float temp0 = (eta2 - s2) - k2;
    const float a2pb2 = __builtin_sqrtf(fmax_c(temp0 * temp0 + (4.0f * k2) * eta2, 0.0f));
    const float temp1 = a2pb2 + c2;
    const float a = __builtin_sqrtf(fmax_c((a2pb2 + temp0) * 0.5f, 0.0f));
    const float temp2 = (2.0f * a) * cosI;
    const float Rs2 = fdiv(temp1 - temp2, temp1 + temp2);
    const float temp3 = a2pb2 * c2 + s2 * s2;
    const float temp4 = temp2 * s2;
    const float Rp2 = fdiv(Rs2 * (temp3 - temp4), temp3 + temp4);
    return 0.5f * (Rp2 + Rs2);
}
// GGX :180-184, denominator associated as (PI*ax) * ((ay*l2)*l2)
DEV float ggx(float hx, float hy, float hz, float ax, float ay) {
    const float sx = fdiv(hx, ax), sy = fdiv(hy, ay);
    const float l2 = (hz * hz + sy * sy) + sx * sx;
    return frcp((PT_PI * ax) * ((ay * l2) * l2));
}

struct Mat {
    float4 m0, m1, m2;  // {emission, type} {param0, alpha.x} {param1, alpha.y}
};
DEV Mat load_mat(const DevScene &sc, const float4 *lds_mats, int m) {
    Mat r;
    if (sc.mats_in_lds) {
        const lds_cf4 q = (lds_cf4)lds_mats + 3 * m;
        r.m0 = to_f4(q[0]); r.m1 = to_f4(q[1]); r.m2 = to_f4(q[2]);
    } else {
        const glb_cf4 q = (glb_cf4)sc.mats + 3 * m;
        r.m0 = to_f4(q[0]); r.m1 = to_f4(q[1]); r.m2 = to_f4(q[2]);
    }
    return r;
}

// ------------------------------------------------------------------------------------------ radiance :409-559
// Path state between bounces.  The reference's depth loop (:416-556) is split so that one call to
// bounce() executes one iteration of it; that lets the persistent kernel keep lanes at different
// depths (and of different pixels) side by side in one wavefront.
struct Path {
    float ox, oy, oz, dx, dy, dz;  // current ray
    float bx, by, bz;              // beta (path throughput)
    float Lx, Ly, Lz;              // radiance gathered so far
    int depth;
    bool spec;                     // extension kernel only: the previous bounce was a specular (dielectric) one
};

DEV void path_begin(Path &P, float ox, float oy, float oz, float dx, float dy, float dz) {
    P.ox = ox; P.oy = oy; P.oz = oz; P.dx = dx; P.dy = dy; P.dz = dz;
    P.bx = P.by = P.bz = 1.f;
    P.Lx = P.Ly = P.Lz = 0.f;
    P.depth = 0;
    P.spec = false;
}

// Everything one iteration of the depth loop does EXCEPT its two BVH traversals: given the closest hit
// `h` of the current ray it applies emission, samples the BSDF, prepares next-event estimation (the
// shadow ray and both possible values of L: light sample accepted or rejected), advances the ray /
// beta / depth and plays Russian roulette.  None of that depends on the shadow ray's result: in the
// reference the acceptance test (:367) only selects whether e*f*G/pdf or 0 is added (:539), and every
// rand() call is made regardless.  The megakernels and the wavefront pipeline share this one copy of
// the shading arithmetic.  On return P holds the next ray / beta / depth; P.L is NOT updated: the
// caller sets it to Lpass or Lfail.
struct Shade {
    bool ended;           // the depth loop ends after this iteration (break, depth limit, roulette)
    bool has_shadow;      // a shadow ray must be traced from P.o along sd: L = accepted ? Lpass : Lfail
    float sdx, sdy, sdz;  // shadow ray direction
    float dist;           // |p - x| for the acceptance test |dist - tHit| < EPS
    float Lpx, Lpy, Lpz;  // L with the light sample accepted (== L itself when !has_shadow)
    float Lfx, Lfy, Lfz;  // L with it rejected
    bool untraced;        // the reference traces a shadow ray here, but both outcomes give the same L bit for bit (no
                          // contribution: a cosine <= 0, or one too small to register): counted as a ray, not traced
};

// Surface at a hit: shading normal and material.  surf_tri() is the reference's (:254 and the triangle's material id);
// the extension kernel also produces it for analytic spheres.
struct Surf {
    float nx, ny, nz;
    int mtrl;
};
DEV Surf surf_tri(const DevScene &sc, const Hit &h) {
    // normal of the closest hit (:254), computed once instead of per candidate
    // (the material id is kept in the unused fourth word of the first normal -- glrtx.hip: pack_scene -- : three gathers per shaded hit instead of four)
    const float4 N0 = sc.nrms[3 * h.tri], N1 = sc.nrms[3 * h.tri + 1], N2 = sc.nrms[3 * h.tri + 2];
    const float w0 = (1.0f - h.u) - h.v;
    const float tx = (w0 * N0.x + h.u * N1.x) + h.v * N2.x;
    const float ty = (w0 * N0.y + h.u * N1.y) + h.v * N2.y;
    const float tz = (w0 * N0.z + h.u * N1.z) + h.v * N2.z;
    const float r = rsq(dot3(tx, ty, tz, tx, ty, tz));
    Surf S;
    S.nx = tx * r; S.ny = ty * r; S.nz = tz * r;
    S.mtrl = __float_as_int(N0.w);
    return S;
}

// Extensions beyond the reference (PARITY UNPINNED -- the reference has neither, so there is nothing to
// compare with; checked against this build's own CPU restatement and against the tessellation limit of the pinned path):
constexpr int EXT_DIELECTRIC = 1;  // materials of type MTRL_DIELECTRIC (raytrace.frag:32, never branched on there: black) reflect / refract
constexpr int EXT_WHITTED = 2;     // Whitted-style: a diffuse surface gathers its direct light and the path ends there; only specular bounces continue

template <bool EXT>
DEV void shade_core(const KernelArgs &a, const float4 *lds_mats, Rng &rng, Path &P, float h_t, bool h_hit, const Surf &S, int ext_flags, Shade &out) {
    const DevScene &sc = a.sc;
    float ox = P.ox, oy = P.oy, oz = P.oz, dx = P.dx, dy = P.dy, dz = P.dz;
    float bx = P.bx, by = P.by, bz = P.bz;
    float Lx = P.Lx, Ly = P.Ly, Lz = P.Lz;
    const int depth = P.depth;
    const float nLf = (float)sc.n_light;
    bool done = true;  // every `break` of the reference loop leaves this set
    bool has_shadow = false;
    float sdx = 0.f, sdy = 0.f, sdz = 0.f, sdist = 0.f;
    float px_ = 0.f, py_ = 0.f, pz_ = 0.f;  // beta * candidate contribution
    float zx_ = 0.f, zy_ = 0.f, zz_ = 0.f;  // beta * 0

    bool spec_out = false, stop_after = false;
    do {
        if (!h_hit) break;  // miss: nothing is added and the loop ends (:497-499)
        const float nx = S.nx, ny = S.ny, nz = S.nz;
        const Mat M = load_mat(sc, lds_mats, S.mtrl);
        const int type = __float_as_int(M.m0.w);

        // :420
        const float tt = h_t + PT_EPS;
        const float xx = ox + tt * dx, xy = oy + tt * dy, xz = oz + tt * dz;
        const float woz = (-(dz * nz) - (dy * ny)) - (dx * nx);  // dot(-d, n), also woLocal.z

        if (type == 5 && woz >= PT_EPS) {
            // MTRL_MEDIA from the front: the volume branch is compiled out in the reference
            // (ENABLE_VOLUME 0, :424-487); the ray is left unchanged.
        } else if (EXT && type == 4 && (ext_flags & EXT_DIELECTRIC)) {
            // ---- extension: smooth dielectric, param0 = tint, param1.x = index of refraction.  One rand() picks the reflected
            // or the refracted direction with the unpolarised Fresnel reflectance as probability (weight = tint either way);
            // delta BSDF: no light sampling; emission met on the next bounce counts (the "specularReflect" the reference
            // declares at :490 and never sets).
            if (depth == 0 || P.spec) { Lx = Lx + bx * M.m0.x; Ly = Ly + by * M.m0.y; Lz = Lz + bz * M.m0.z; }
            const float rl = rsq(dot3(dx, dy, dz, dx, dy, dz));  // directions are not renormalised along a path (:542)
            const float ux = dx * rl, uy = dy * rl, uz = dz * rl;
            const float ci0 = (-(uz * nz) - (uy * ny)) - (ux * nx);  // cos(incident, n); negative: leaving the medium
            const bool entering = ci0 > 0.0f;
            const float fnx = entering ? nx : -nx, fny = entering ? ny : -ny, fnz = entering ? nz : -nz;
            const float ci = __builtin_fabsf(ci0);
            const float ior = M.m2.x;
            const float eta = entering ? frcp(ior) : ior;
            const float k = 1.0f - (eta * eta) * (1.0f - ci * ci);
            float F = 1.0f;  // total internal reflection
            float ct = 0.0f;
            if (k > 0.0f) {
                ct = __builtin_sqrtf(k);
                const float rs = (eta * ci - ct) / (eta * ci + ct);
                const float rp = (ci - eta * ct) / (ci + eta * ct);
                F = 0.5f * (rs * rs + rp * rp);
            }
            const float pick = pt_rand(rng);
            const float hx = ox + h_t * dx, hy = oy + h_t * dy, hz = oz + h_t * dz;  // the hit point itself
            if (pick < F) {  // reflect
                const float two = 2.0f * ci;
                dx = ux + two * fnx; dy = uy + two * fny; dz = uz + two * fnz;
                ox = hx + fnx * (2.0f * PT_EPS); oy = hy + fny * (2.0f * PT_EPS); oz = hz + fnz * (2.0f * PT_EPS);
            } else {  // refract
                const float g = eta * ci - ct;
                dx = eta * ux + g * fnx; dy = eta * uy + g * fny; dz = eta * uz + g * fnz;
                ox = hx - fnx * (2.0f * PT_EPS); oy = hy - fny * (2.0f * PT_EPS); oz = hz - fnz * (2.0f * PT_EPS);
            }
            bx = bx * M.m1.x; by = by * M.m1.y; bz = bz * M.m1.z;
            spec_out = true;
        } else {
            if (depth == 0 || (EXT && P.spec)) {  // :490-494 (specularReflect / passedVolume are never set by the reference)
                Lx = Lx + bx * M.m0.x; Ly = Ly + by * M.m0.y; Lz = Lz + bz * M.m0.z;
            }
            // :502-506 local frame; cross() with the selected axis kept as 0/1 multipliers
            const float B = (0.1f < __builtin_fabsf(nx)) ? 1.0f : 0.0f, A = 1.0f - B;
            const float ux = B * nz;
            const float nuy = A * nz;  // = -u.y
            const float uz = A * ny - B * nx;
            const float vx = ny * uz + nuy * nz;
            const float vy = nz * ux - nx * uz;
            const float vz = -(nuy * nx) - (ny * ux);
            const float wox = (-(dz * uz) + nuy * dy) - (dx * ux);
            const float woy = (-(dz * vz) - (dy * vy)) - (dx * vx);

            float fx = 0.f, fy = 0.f, fz = 0.f, pdf = 1.0f;
            float wlx = 0.f, wly = 0.f, wlz = 1.0f;
            if (type == 2) {
                // diffuse :511-519
                const float ra = pt_rand(rng);
                const float rb = pt_rand(rng);
                const float r1 = PT_2PI * ra;
                const float r2s = __builtin_sqrtf(rb);
                wlx = pt_cos(r1) * r2s;
                wly = pt_sin(r1) * r2s;
                wlz = __builtin_sqrtf(1.0f - rb);
                if (sc.mats_in_lds) { fx = M.m2.x; fy = M.m2.y; fz = M.m2.z; }  // albedo / PI, formed when the materials were staged (stage_mats)
                else { fx = div_pi(M.m1.x); fy = div_pi(M.m1.y); fz = div_pi(M.m1.z); }
                pdf = div_pi(wlz);
            } else if (type == 3) {
                // conductor :520-532; param0 = kappa, param1 = eta
                const float ax = M.m1.w, ay = M.m2.w;
                const float u0 = pt_rand(rng);
                const float u1 = pt_rand(rng);
                // sampleGGXVNDF :195-214
                const float sx = wox * ax, sy = woy * ay;
                const float lw = (woz * woz + sy * sy) + sx * sx;
                const float rw = rsq(lw);
                const float vhx = sx * rw, vhy = sy * rw, vhz = woz * rw;
                const float lensq = vhx * vhx + vhy * vhy;
                const float q = rsq(lensq);
                const float T1x = (0.0f < lensq) ? -(vhy * q) : 1.0f;
                const float T1y = (0.0f < lensq) ? vhx * q : 0.0f;
                const float rr = __builtin_sqrtf(u0);
                const float phi = PT_2PI * u1;
                const float t1 = rr * pt_cos(phi);
                const float t2r = rr * pt_sin(phi);
                const float s = 0.5f * (1.0f + vhz);
                const float c1 = 1.0f - t1 * t1;
                const float t2 = (1.0f - s) * __builtin_sqrtf(c1) + s * t2r;
                const float T2y = vhz * T1x;
                const float zq = vhz * T1y;  // = -T2.x
                const float T2z = vhx * T1y - vhy * T1x;
                float nhx = t1 * T1x - zq * t2;
                float nhy = t1 * T1y + t2 * T2y;
                float nhz = t2 * T2z;
                const float sq2 = __builtin_sqrtf(fmax_c(c1 - t2 * t2, 0.0f));
                nhx = nhx + sq2 * vhx; nhy = nhy + sq2 * vhy; nhz = nhz + sq2 * vhz;
                const float nex = nhx * ax, ney = nhy * ay, nez = fmax_c(nhz, 0.0f);
                const float rn = rsq((nez * nez + ney * ney) + nex * nex);
                const float whx = nex * rn, why = ney * rn, whz = nez * rn;
                // wiLocal = 2 dot(wh, wo) wh - wo :527
                const float dwh = (whz * woz + why * woy) + whx * wox;
                const float two = 2.0f * dwh;
                const float hx2 = two * whx, hy2 = two * why, hz2 = two * whz;  // = wi + wo
                wlx = hx2 - wox; wly = hy2 - woy; wlz = hz2 - woz;
                const float c2 = wlz * wlz, s2 = 1.0f - c2;
                const float Fx = fresnel1(c2, s2, wlz, M.m2.x, M.m1.x);
                const float Fy = fresnel1(c2, s2, wlz, M.m2.y, M.m1.y);
                const float Fz = fresnel1(c2, s2, wlz, M.m2.z, M.m1.z);
                // microfacetGGXBRDF :186-193
                const float rh = rsq((hz2 * hz2 + hy2 * hy2) + hx2 * hx2);
                const float D = ggx(hx2 * rh, hy2 * rh, hz2 * rh, ax, ay);
                const float wisx = wlx * ax, wisy = wly * ay;
                const float len_wi = __builtin_sqrtf((c2 + wisy * wisy) + wisx * wisx);
                const float len_wo = __builtin_sqrtf(lw);
                const float den = 2.0f * (__builtin_fabsf(woz) * len_wi + __builtin_fabsf(wlz) * len_wo);
                const float brdf = fdiv(D, den);
                fx = Fx * brdf; fy = Fy * brdf; fz = Fz * brdf;
                // weightedGGXPDF :216-219
                const float D2 = ggx(whx, why, whz, ax, ay);
                const float g1 = fdiv(0.5f, len_wo + woz);
                const float pn = (g1 * D2) * fmax_c(dwh, 0.0f);
                const float dwi = (wlz * whz + wly * why) + wlx * whx;
                pdf = fdiv(pn, fmax_c(dwi, PT_EPS));
            }

            // isBlack(f) || pdf == 0 :534, evaluated as min(|f|, |pdf|) == 0
            {
                const float lf = __builtin_sqrtf((fz * fz + fy * fy) + fx * fx);
                if (fmin_g(lf, __builtin_fabsf(pdf)) == 0.0f) break;
            }

            // ---- sampleDirect :337-403
            float cx = 0.f, cy = 0.f, cz = 0.f;
            const float sox = xx + nx * PT_EPS, soy = xy + ny * PT_EPS, soz = xz + nz * PT_EPS;  // spawnRay :121-123
            {
                const float rl = pt_rand(rng);
                int lid = (int)(rl * nLf);
                lid = (sc.n_light - 1 < lid) ? sc.n_light - 1 : lid;
                const float ua0 = pt_rand(rng);
                const float ub0 = pt_rand(rng);
                const bool flip = 1.0f < ua0 + ub0;
                const float ua = flip ? 1.0f - ua0 : ua0;
                const float ub = flip ? 1.0f - ub0 : ub0;
                const float w0 = (1.0f - ua) - ub;
                float4 V0, V1, V2, N0, N1, N2;
                if (lid >= 0 && sc.lights_in_lds) {  // (wave-uniform) staged behind the materials: six LDS reads instead of six gathers
                    const lds_cf4 q = (lds_cf4)lds_mats + (sc.mats_in_lds ? 3 * sc.n_mat : 0) + 6 * lid;
                    V0 = to_f4(q[0]); V1 = to_f4(q[1]); V2 = to_f4(q[2]); N0 = to_f4(q[3]); N1 = to_f4(q[4]); N2 = to_f4(q[5]);
                } else if (lid >= 0) {
                    const float4 *Lp = sc.lights + 6 * lid;
                    V0 = Lp[0]; V1 = Lp[1]; V2 = Lp[2]; N0 = Lp[3]; N1 = Lp[4]; N2 = Lp[5];
                } else {  // u_nLights == 0: out-of-range texelFetch returns zeros in the reference's GL
                    V0 = V1 = V2 = N0 = N1 = N2 = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                const float px = (w0 * V0.x + ua * V1.x) + ub * V2.x;
                const float py = (w0 * V0.y + ua * V1.y) + ub * V2.y;
                const float pz = (w0 * V0.z + ua * V1.z) + ub * V2.z;
                const float nlx = (w0 * N0.x + ua * N1.x) + ub * N2.x;
                const float nly = (w0 * N0.y + ua * N1.y) + ub * N2.y;
                const float nlz = (w0 * N0.z + ua * N1.z) + ub * N2.z;
                const float dvx = px - xx, dvy = py - xy, dvz = pz - xz;
                const float dd = (dvz * dvz + dvy * dvy) + dvx * dvx;
                const float rd = rsq(dd);
                const float dirx = dvx * rd, diry = dvy * rd, dirz = dvz * rd;
                has_shadow = true;
                sdx = dirx; sdy = diry; sdz = dirz;
                sdist = __builtin_sqrtf(dd);
                {  // the contribution IF the shadow ray passes the test at :367
                    float gx = 0.f, gy = 0.f, gz = 0.f;
                    if (type == 2) {
                        gx = M.m1.x; gy = M.m1.y; gz = M.m1.z;  // albedo without 1/PI :372
                    } else if (type == 3) {
                        const float ax = M.m1.w, ay = M.m2.w;
                        const float cosI = fmax_c((-(dirz * nz) - (diry * ny)) - (dirx * nx), 0.0f);
                        const float c2 = cosI * cosI, s2 = 1.0f - c2;
                        const float Fx = fresnel1(c2, s2, cosI, M.m2.x, M.m1.x);
                        const float Fy = fresnel1(c2, s2, cosI, M.m2.y, M.m1.y);
                        const float Fz = fresnel1(c2, s2, cosI, M.m2.z, M.m1.z);
                        const float ilx = (uz * dirz - nuy * diry) + ux * dirx;
                        const float ily = (vz * dirz + vy * diry) + vx * dirx;
                        const float ilz = (dirz * nz + diry * ny) + dirx * nx;
                        const float hx = ilx + wox, hy = ily + woy, hz = ilz + woz;
                        const float rh = rsq((hz * hz + hy * hy) + hx * hx);
                        const float D = ggx(hx * rh, hy * rh, hz * rh, ax, ay);
                        const float wisx = ilx * ax, wisy = ily * ay;
                        const float len_wi = __builtin_sqrtf((ilz * ilz + wisy * wisy) + wisx * wisx);
                        const float wosx = wox * ax, wosy = woy * ay;
                        const float len_wo = __builtin_sqrtf((woz * woz + wosy * wosy) + wosx * wosx);
                        const float den = 2.0f * (__builtin_fabsf(woz) * len_wi + __builtin_fabsf(ilz) * len_wo);
                        const float brdf = fdiv(D, den);
                        gx = Fx * brdf; gy = Fy * brdf; gz = Fz * brdf;
                    }
                    const Mat LM = load_mat(sc, lds_mats, __float_as_int(V0.w));
                    const float dot0 = (dirz * nz + diry * ny) + dirx * nx;
                    const float dot1 = (-(dirz * nlz) - (diry * nly)) - (dirx * nlx);
                    if (0.0f < fmin_g(dot0, dot1)) {
                        const float e1x = V1.x - V0.x, e1y = V1.y - V0.y, e1z = V1.z - V0.z;
                        const float e2x = V2.x - V0.x, e2y = V2.y - V0.y, e2z = V2.z - V0.z;
                        const float kx = e1y * e2z - e1z * e2y;
                        const float ky = e1z * e2x - e1x * e2z;
                        const float kz = e1x * e2y - e1y * e2x;
                        const float G = fdiv(dot0 * dot1, dd);  // dist*dist is folded to dd
                        const float area = 0.5f * __builtin_sqrtf((kz * kz + ky * ky) + kx * kx);
                        const float lpdf = frcp(area * nLf);
                        cx = fdiv((LM.m0.x * gx) * G, lpdf);
                        cy = fdiv((LM.m0.y * gy) * G, lpdf);
                        cz = fdiv((LM.m0.z * gz) * G, lpdf);
                    }
                }
            }
            // :539 L += beta * sampleDirect(): both outcomes of the acceptance test
            px_ = bx * cx; py_ = by * cy; pz_ = bz * cz;
            zx_ = bx * 0.0f; zy_ = by * 0.0f; zz_ = bz * 0.0f;
            // :542-544; wi is not renormalised
            const float wix = (ux * wlx + vx * wly) + nx * wlz;
            const float wiy = (-(nuy * wlx) + vy * wly) + ny * wlz;
            const float wiz = (uz * wlx + vz * wly) + nz * wlz;
            ox = sox; oy = soy; oz = soz;
            dx = wix; dy = wiy; dz = wiz;
            const float cw = fmax_c((nz * wiz + ny * wiy) + nx * wix, 0.0f);
            bx = bx * fdiv(fx * cw, pdf);
            by = by * fdiv(fy * cw, pdf);
            bz = bz * fdiv(fz * cw, pdf);
            if (EXT && (ext_flags & EXT_WHITTED) && type == 2) stop_after = true;  // Whitted: direct light only at a diffuse surface
        }
        if (EXT && stop_after) break;

        // Russian roulette :549-555
        if (2 < depth) {
            float pm = fmax_g(by, bz);
            pm = fmax_g(bx, pm);
            const float pq = fmin_c(pm, 0.95f);
            const float rr = pt_rand(rng);
            if (pq < rr) break;
            bx = fdiv(bx, pq); by = fdiv(by, pq); bz = fdiv(bz, pq);
        }
        done = false;
    } while (false);

    P.ox = ox; P.oy = oy; P.oz = oz; P.dx = dx; P.dy = dy; P.dz = dz;
    P.bx = bx; P.by = by; P.bz = bz;
    P.depth = depth + 1;
    if (EXT) P.spec = spec_out;
    out.ended = done || P.depth >= a.max_depth;
    out.has_shadow = has_shadow;
    out.untraced = false;
    out.sdx = sdx; out.sdy = sdy; out.sdz = sdz; out.dist = sdist;
    if (has_shadow) {
        out.Lpx = Lx + px_; out.Lpy = Ly + py_; out.Lpz = Lz + pz_;
        out.Lfx = Lx + zx_; out.Lfy = Ly + zy_; out.Lfz = Lz + zz_;
        if (__float_as_uint(out.Lpx) == __float_as_uint(out.Lfx) && __float_as_uint(out.Lpy) == __float_as_uint(out.Lfy) &&
            __float_as_uint(out.Lpz) == __float_as_uint(out.Lfz)) {
            out.has_shadow = false;  // whatever the shadow ray finds, L is the same
            out.untraced = true;
        }
    } else {
        out.Lpx = out.Lfx = Lx; out.Lpy = out.Lfy = Ly; out.Lpz = out.Lfz = Lz;
    }
}

// The reference's iteration: triangles only, no extensions.
DEV void shade_hit(const KernelArgs &a, const float4 *lds_mats, Rng &rng, Path &P, const Hit &h, Shade &out) {
    Surf S;
    S.nx = S.ny = S.nz = 0.f; S.mtrl = 0;
    if (h.tri >= 0) S = surf_tri(a.sc, h);
    shade_core<false>(a, lds_mats, rng, P, h.t, h.tri >= 0, S, 0, out);
}

// The acceptance test of sampleDirect (:367): the shadow ray hit something at the light sample's distance.
DEV bool nee_accepted(float dist, float tS, bool hit) { return hit && __builtin_fabsf(dist - tS) < PT_EPS; }

// One iteration of the depth loop (megakernel form).  Returns true when the loop ends (a `break`, or
// depth reaching u_maxDepth).  Precondition: P.depth < a.max_depth.
DEV bool bounce(const KernelArgs &a, const float4 *lds_mats, int *stack, Rng &rng, Path &P, unsigned long long &rays) {
    const Hit h = traverse<true>(a.sc, stack, P.ox, P.oy, P.oz, P.dx, P.dy, P.dz);
    rays++;
    Shade sh;
    shade_hit(a, lds_mats, rng, P, h, sh);
    bool ok = false;
    if (sh.has_shadow) {
        const Hit s = traverse<false>(a.sc, stack, P.ox, P.oy, P.oz, sh.sdx, sh.sdy, sh.sdz, shadow_limit(sh.dist, a.sc.shadow_limited), sh.dist);
        rays++;
        ok = nee_accepted(sh.dist, s.t, s.tri >= 0);
    } else if (sh.untraced) {
        rays += (1ull << 32) + 1ull;  // an execution of intersect() in the reference; its result cannot change L (Shade::untraced)
    }
    P.Lx = ok ? sh.Lpx : sh.Lfx; P.Ly = ok ? sh.Lpy : sh.Lfy; P.Lz = ok ? sh.Lpz : sh.Lfz;
    return sh.ended;
}

// ------------------------------------------------------------------------------------------ extension kernel (f4)
// Analytic spheres next to the triangle BVH, dielectric materials, Whitted-style termination.  PARITY UNPINNED: the reference
// has none of this (its scenes are triangle meshes, raytrace.frag:226-257; MTRL_DIELECTRIC is a dead constant, :32), so there
// is no reference output to compare with.  The arithmetic the reference does have -- everything in shade_core<false> -- is
// shared, not copied.  Launched only when the caller asks for it (glrtx_upload_spheres / glrtx_set_extensions).
struct ExtArgs {
    const float4 *spheres;  // {centre.xyz, radius}; tested one by one after the BVH (BASELINE's sphere configs have 3-8 of them)
    const int *sphere_mat;  // material id per sphere
    int n_spheres;
    int flags;              // EXT_*
};
constexpr int kMaxSpheres = 1024;

// Closest intersection of a ray with sphere k beyond EPS: |o + t d - c|^2 = r^2 solved with the half-b form, IEEE ops only
// (the CPU restatement in oracle/pt_oracle.c performs the same operations in the same order).
DEV float sphere_t(float4 sp, float ox, float oy, float oz, float dx, float dy, float dz) {
    const float cx = ox - sp.x, cy = oy - sp.y, cz = oz - sp.z;
    const float A = dot3(dx, dy, dz, dx, dy, dz);
    const float B = dot3(cx, cy, cz, dx, dy, dz);
    const float C = dot3(cx, cy, cz, cx, cy, cz) - sp.w * sp.w;
    const float disc = B * B - A * C;
    if (!(disc >= 0.0f)) return PT_INFTY;
    const float sq = __builtin_sqrtf(disc);
    const float t0 = (-B - sq) / A, t1 = (-B + sq) / A;
    const float t = t0 > PT_EPS ? t0 : t1;
    return t > PT_EPS ? t : PT_INFTY;
}

// Closest hit over triangles (the reference traversal) and spheres; Hit::tri >= 0 triangle, -2 - k sphere k, -1 miss.
// The spheres are staged into LDS at kernel start (ext_stage_spheres): the scan reads them with a wave-uniform ds_read_b128 each --
// "scene primitives staged into LDS" as BASELINE.json's north_star words it.
template <bool CLOSEST>
DEV Hit traverse_ext(const DevScene &sc, const ExtArgs &ex, const float4 *lds_spheres, int *stack, float ox, float oy, float oz, float dx, float dy, float dz,
                     float limit = PT_INFTY, float stop_d = -__builtin_inff()) {
    Hit h = traverse<CLOSEST>(sc, stack, ox, oy, oz, dx, dy, dz, limit, stop_d);
    const lds_cf4 q = (lds_cf4)lds_spheres;
    for (int k = 0; k < ex.n_spheres; k++) {
        const float t = sphere_t(to_f4(q[k]), ox, oy, oz, dx, dy, dz);
        if (t < h.t) { h.t = t; h.tri = -2 - k; }
    }
    return h;
}

DEV bool bounce_ext(const KernelArgs &a, const ExtArgs &ex, const float4 *lds_spheres, const float4 *lds_mats, int *stack, Rng &rng, Path &P, unsigned long long &rays) {
    const Hit h = traverse_ext<true>(a.sc, ex, lds_spheres, stack, P.ox, P.oy, P.oz, P.dx, P.dy, P.dz);
    rays++;
    Surf S;
    S.nx = S.ny = S.nz = 0.f; S.mtrl = 0;
    if (h.tri >= 0) S = surf_tri(a.sc, h);
    else if (h.tri < -1) {
        const float4 sp = to_f4(((lds_cf4)lds_spheres)[-2 - h.tri]);
        const float qx = (P.ox + h.t * P.dx) - sp.x, qy = (P.oy + h.t * P.dy) - sp.y, qz = (P.oz + h.t * P.dz) - sp.z;
        const float r = rsq(dot3(qx, qy, qz, qx, qy, qz));
        S.nx = qx * r; S.ny = qy * r; S.nz = qz * r;
        S.mtrl = ex.sphere_mat[-2 - h.tri];
    }
    Shade sh;
    shade_core<true>(a, lds_mats, rng, P, h.t, h.tri != -1, S, ex.flags, sh);
    bool ok = false;
    if (sh.has_shadow) {
        const Hit s = traverse_ext<false>(a.sc, ex, lds_spheres, stack, P.ox, P.oy, P.oz, sh.sdx, sh.sdy, sh.sdz, shadow_limit(sh.dist, a.sc.shadow_limited), sh.dist);
        rays++;
        ok = nee_accepted(sh.dist, s.t, s.tri != -1);
    } else if (sh.untraced) {
        rays += (1ull << 32) + 1ull;
    }
    P.Lx = ok ? sh.Lpx : sh.Lfx; P.Ly = ok ? sh.Lpy : sh.Lfy; P.Lz = ok ? sh.Lpz : sh.Lfz;
    return sh.ended;
}

// primary ray of one sample, main() :577-607
// `cam` = the launch's camera block {c2w[16], s2c[16], aperture, focal}: a.cam, or the
// workgroup's LDS copy of it.
constexpr int kCamFloats = 34;  // == sizeof(KernelArgs::cam) / 4
DEV void camera_ray(const KernelArgs &a, const float *cam, Rng &rng, float fcx, float fcy, Path &P) {
    const float W = (float)a.width, H = (float)a.height;
    const float *S = cam + 16, *C = cam;
    const float aperture = cam[32], focal = cam[33];
    const float r0 = pt_rand(rng);
    const float r1 = pt_rand(rng);
    const float nx = ((fcx + r0) / W) * 2.0f + -1.0f;
    const float ny = ((fcy + r1) / H) * 2.0f + -1.0f;
    // u_s2cMat * (nx, ny, 0, 1), summed as (col0*nx + col3) + col1*ny
    const float tx = (S[0] * nx + S[12]) + S[4] * ny;
    const float ty = (S[1] * nx + S[13]) + S[5] * ny;
    const float tz = (S[2] * nx + S[14]) + S[6] * ny;
    const float tw = (S[3] * nx + S[15]) + S[7] * ny;
    const float cx = tx / tw, cy = ty / tw, cz = tz / tw;
    const float rn = rsq((cz * cz + cy * cy) + cx * cx);
    float dx = cx * rn, dy = cy * rn, dz = cz * rn;
    float lox = 0.0f, loy = 0.0f;
    if (0.0f < aperture) {  // thin lens :589-598
        const float ra = pt_rand(rng);
        const float rb = pt_rand(rng);
        const float r = __builtin_sqrtf(ra) * aperture;
        const float th = PT_2PI * rb;
        lox = r * pt_cos(th);
        loy = r * pt_sin(th);
        const float ft = (-focal) / dz;
        const float fx = dx * ft - lox, fy = dy * ft - loy, fz = dz * ft;
        const float rf = rsq((fz * fz + fy * fy) + fx * fx);
        dx = fx * rf; dy = fy * rf; dz = fz * rf;
    }
    // u_c2wMat * (o, 1), divided by w; u_c2wMat * (d, 0), normalised (:601-607)
    const float wx = (C[0] * lox + C[12]) + C[4] * loy;
    const float wy = (C[1] * lox + C[13]) + C[5] * loy;
    const float wz = (C[2] * lox + C[14]) + C[6] * loy;
    const float ww = (C[3] * lox + C[15]) + C[7] * loy;
    const float ex = (C[0] * dx + C[4] * dy) + C[8] * dz;
    const float ey = (C[1] * dx + C[5] * dy) + C[9] * dz;
    const float ez = (C[2] * dx + C[6] * dy) + C[10] * dz;
    const float re = rsq((ez * ez + ey * ey) + ex * ex);
    path_begin(P, wx / ww, wy / ww, wz / ww, ex * re, ey * re, ez * re);
}

// Blocks are dealt round-robin to the 8 XCDs; give each XCD a contiguous run of screen tiles so
// that neighbouring tiles (which walk the same BVH subtrees) share one L2.  Bijective for any n.
DEV int xcd_swizzle(int bid, int n) {
    constexpr int X = 8;
    const int per = n / X, rem = n % X;
    const int xcd = bid % X, k = bid / X;
    // XCD x owns per + (x < rem) tiles; tiles of XCD x start at x*per + min(x, rem)
    return xcd * per + (xcd < rem ? xcd : rem) + k;
}

// ------------------------------------------------------------------------------------------ main :565-614
DEV int local_row_to_y(const KernelArgs &a, int lrow) {  // owned stripe s holds global stripe s*world + rank
    return ((lrow / a.stripe) * a.world + a.rank) * a.stripe + lrow % a.stripe;
}

// Materials into LDS (at most kMaxLdsMaterials).  A diffuse material's BSDF value -- albedo / PI (:514), the same three quotients at every diffuse hit -- is
// formed here, once per workgroup, in the slot a diffuse material leaves unused (param1); shade_core reads it instead of dividing.
DEV void stage_mats(const KernelArgs &a, float4 *lds_mats) {
    for (int m = threadIdx.x; m < a.sc.n_mat; m += kBlockThreads) {
        const float4 m0 = a.sc.mats[3 * m], m1 = a.sc.mats[3 * m + 1];
        float4 m2 = a.sc.mats[3 * m + 2];
        if (__float_as_int(m0.w) == 2) { m2.x = m1.x / PT_PI; m2.y = m1.y / PT_PI; m2.z = m1.z / PT_PI; }
        lds_mats[3 * m] = m0; lds_mats[3 * m + 1] = m1; lds_mats[3 * m + 2] = m2;
    }
}
// The light triangles behind them (DevScene::lights_in_lds).
DEV void stage_lights(const KernelArgs &a, float4 *lds_mats) {
    if (!a.sc.lights_in_lds) return;
    float4 *dst = lds_mats + (a.sc.mats_in_lds ? 3 * a.sc.n_mat : 0);
    for (int i = threadIdx.x; i < 6 * a.sc.n_light; i += kBlockThreads) dst[i] = a.sc.lights[i];
}

DEV void lds_setup(const KernelArgs &a, unsigned char *lds_raw, float4 *&lds_mats, int *&stack) {
    lds_mats = reinterpret_cast<float4 *>(lds_raw);
    const int mat_f4 = a.sc.lds_head_f4;  // materials, then lights
    stack = reinterpret_cast<int *>(lds_raw + (size_t)mat_f4 * sizeof(float4)) + 2 * threadIdx.x;  // 8-byte entries
    if (a.sc.mats_in_lds) stage_mats(a, lds_mats);
    stage_lights(a, lds_mats);
    if (a.sc.lds_head_f4 > 0) __syncthreads();
}

// Extension kernel: the analytic spheres ({centre, radius}, 16 B each, at most kMaxSpheres) go into LDS behind the traversal stacks.
DEV float4 *ext_stage_spheres(const KernelArgs &a, const ExtArgs &ex, unsigned char *lds_raw) {
    const int mat_f4 = a.sc.lds_head_f4;  // materials, then lights
    float4 *dst = reinterpret_cast<float4 *>(lds_raw + (size_t)mat_f4 * sizeof(float4) + (size_t)2 * a.sc.stack_entries * kBlockThreads * sizeof(int));
    for (int i = threadIdx.x; i < ex.n_spheres; i += kBlockThreads) dst[i] = ex.spheres[i];
    __syncthreads();
    return dst;
}

template <bool COUNT_RAYS>
DEV void flush_rays(const KernelArgs &a, unsigned long long rays) {
    if (COUNT_RAYS) {  // wave-level sums, one atomic per wavefront and counter
        unsigned long long r = rays & 0xFFFFFFFFull, u = rays >> 32;
        for (int off = 32; off > 0; off >>= 1) { r += __shfl_down(r, off); u += __shfl_down(u, off); }
        if ((threadIdx.x & 63) == 0) {
            if (r) atomicAdd(a.ray_counter, r);
            if (u) atomicAdd(a.ray_counter + 1, u);
        }
    }
}

// Variant A ("tile"): one work-item per pixel, a 16x16 tile per workgroup, the whole sample/depth loop
// nest in one lane.  Simple, but lanes whose path ended early idle until the longest path of the wave
// is done (measured SIMD lane utilisation ~12 % on the headline config, profiles/r01a).
template <bool COUNT_RAYS>
__global__ __launch_bounds__(kBlockThreads) void pt_render_kernel(const KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float4 *lds_mats;
    int *stack;
    lds_setup(a, lds_raw, lds_mats, stack);

    const int tile = xcd_swizzle(blockIdx.x, a.n_tiles);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lx = (tile % a.tiles_x) * kTile + (wave & 1) * 8 + (lane & 7);
    const int lrow = (tile / a.tiles_x) * kTile + (wave >> 1) * 8 + (lane >> 3);
    unsigned long long rays = 0;  // low half: reference rays, high half: those resolved without a traversal
    if (lx < a.width && lrow < a.owned_rows) {
        const int gy = local_row_to_y(a, lrow);
        const float fcx = (float)lx + 0.5f, fcy = (float)gy + 0.5f;  // gl_FragCoord.xy
        Rng rng;
        rng.x = fcx / (float)a.width; rng.y = fcy / (float)a.height; rng.sx = a.seed_x; rng.sy = a.seed_y;  // :567
        float4 *px = a.accum + (size_t)lrow * a.pitch_f4 + lx;
        float4 acc = *px;  // previous (L, count): read-modify-write replaces the ping-pong FBOs (:570-572)
        for (int i = 0; i < a.n_samples; i++) {
            Path P;
            camera_ray(a, a.cam, rng, fcx, fcy, P);
            if (a.max_depth > 0)
                while (!bounce(a, lds_mats, stack, rng, P, rays)) {}
            acc.x = acc.x + fmin_c(P.Lx, 100.0f);  // :558, :608
            acc.y = acc.y + fmin_c(P.Ly, 100.0f);
            acc.z = acc.z + fmin_c(P.Lz, 100.0f);
            acc.w = acc.w + 1.0f;
        }
        *px = acc;  // 8 lanes x 16 B = one 128 B segment per tile row
    }
    flush_rays<COUNT_RAYS>(a, rays);
}

// Variant B ("persistent"): wavefronts stay resident and pull pixels from a global work counter in
// chunks of kChunk; a lane whose pixel is finished takes the next pixel of its wave's chunk at once
// (path regeneration), so lanes at different depths -- and of different pixels -- run side by side
// and a wave's SIMD lanes stay occupied until the image runs out.  Per-pixel arithmetic, RNG stream
// and accumulation order are untouched, so results are bit-identical to variant A.
// Work order: pixel id -> 8x8 tile (row-major over the tile grid) -> pixel within the tile.
constexpr int kChunk = 256;

// The kernel's by-value arguments as they lie in the kernarg segment.  The loop below reads its launch constants from there, through a
// pointer the compiler cannot see through and takes anew in every trip, instead of from the by-value copies: those are loaded once at
// kernel entry and stay live -- in ~100 scalar registers, most of them spilled to vector lanes -- across the traversal loops.
struct PersistKernArgs {
    KernelArgs a;
    unsigned *work_counter;
    ExtArgs ex;
};
DEV const PersistKernArgs *persist_kernargs() {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();  // constant address space
    asm volatile("" : "+s"(p));  // opaque: loads through it stay behind this point
    return (const PersistKernArgs *)p;
}

template <bool COUNT_RAYS, bool EXT = false>
__global__ __launch_bounds__(kBlockThreads) void pt_render_persistent(const KernelArgs a_entry, unsigned *work_counter_entry, const ExtArgs ex_entry) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float4 *lds_mats;
    int *stack;
    (void)work_counter_entry; (void)ex_entry;
    lds_setup(a_entry, lds_raw, lds_mats, stack);
    float4 *lds_spheres = nullptr;
    if (EXT) lds_spheres = ext_stage_spheres(a_entry, ex_entry, lds_raw);
    const KernelArgs &a = persist_kernargs()->a;  // (re-taken inside the loop)

    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int tiles8_x = (a.width + 7) >> 3, tiles8_y = (a.owned_rows + 7) >> 3;
    const int total = tiles8_x * tiles8_y * 64;

    // wave-uniform work state
    int chunk_next = 0, chunk_end = 0;
    bool exhausted = false;
    // per-lane state
    bool alive = false, fresh = false;
    int px_off = 0, sample = 0;
    float fcx = 0.f, fcy = 0.f;
    Rng rng = {0.f, 0.f, a.seed_x, a.seed_y};
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    Path P;
    path_begin(P, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f);
    unsigned long long rays = 0;  // low half: reference rays, high half: those resolved without a traversal

    for (;;) {
        const PersistKernArgs *ka = persist_kernargs();
        const KernelArgs &a = ka->a;
        unsigned *const work_counter = ka->work_counter;
        // ---- regeneration: idle lanes take the next pixels of the wave's chunk
        unsigned long long idle = __ballot(!alive);
        while (idle != 0ull && !exhausted) {
            if (chunk_next >= chunk_end) {
                int base = 0;
                if (lane == 0) base = (int)atomicAdd(work_counter, (unsigned)kChunk);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= total) { exhausted = true; break; }
                chunk_next = base;
                chunk_end = base + kChunk < total ? base + kChunk : total;
            }
            const int n = __popcll(idle);
            const int avail = chunk_end - chunk_next;
            const int take = n < avail ? n : avail;
            const int rank = __popcll(idle & lt_mask);
            if (!alive && rank < take) {
                const int id = chunk_next + rank;
                const int t = id >> 6, w = id & 63;
                const int lx = (t % tiles8_x) * 8 + (w & 7);
                const int lrow = (t / tiles8_x) * 8 + (w >> 3);
                if (lx < a.width && lrow < a.owned_rows && a.n_samples > 0) {
                    const int gy = local_row_to_y(a, lrow);
                    fcx = (float)lx + 0.5f; fcy = (float)gy + 0.5f;  // gl_FragCoord.xy
                    rng.x = fcx / (float)a.width; rng.y = fcy / (float)a.height;  // :567
                    px_off = lrow * a.pitch_f4 + lx;
                    acc = a.accum[px_off];  // previous (L, count) (:570-572)
                    sample = 0;
                    alive = true; fresh = true;
                }
            }
            chunk_next += take;
            // still-idle lanes (chunk ran short, or the pixel drawn lies outside the image) go round again;
            // every round consumes at least one id, so the loop ends when the image is exhausted
            idle = __ballot(!alive);
        }
        if (!__any(alive)) {
            if (exhausted) break;
            continue;
        }
        // ---- one step for every live lane: (new sample ->) one bounce
        if (alive) {
            if (fresh) {
                camera_ray(a, a.cam, rng, fcx, fcy, P);
                fresh = false;
            }
            bool finished = true;
            if (a.max_depth > 0) finished = EXT ? bounce_ext(a, ka->ex, lds_spheres, lds_mats, stack, rng, P, rays) : bounce(a, lds_mats, stack, rng, P, rays);
            if (finished) {
                acc.x = acc.x + fmin_c(P.Lx, 100.0f);  // :558, :608
                acc.y = acc.y + fmin_c(P.Ly, 100.0f);
                acc.z = acc.z + fmin_c(P.Lz, 100.0f);
                acc.w = acc.w + 1.0f;
                sample++;
                if (sample >= a.n_samples) {
                    a.accum[px_off] = acc;
                    alive = false;
                } else {
                    fresh = true;
                }
            }
        }
    }
    flush_rays<COUNT_RAYS>(persist_kernargs()->a, rays);
}

// ------------------------------------------------------------------------------------------ wavefront formulation
// The depth loop cut at its two traversals.  A "trip" is: traverse every queued ray -- the paths'
// next rays AND the shadow rays of the previous bounce (both are closest-hit queries, :363) -- then,
// per live path, resolve the pending shadow ray (:367), run shade_hit() on the new hit and queue the
// next + shadow ray; finished paths add their sample to the accumulator and start the pixel's next
// sample.  The traverse phase pulls RAYS one at a time (a lane whose ray is finished takes the next
// queued ray), so SIMD lanes stay full although traversal length varies 10x between rays; the shade
// phase runs on a compacted list of live paths.  Path state lives in HBM as float4 SoA indexed by
// the pixel's tile-order id.  Per-path arithmetic, RNG call order and the order in which samples
// are added to a pixel are exactly those of the megakernels: results are bit-identical.
// (A device-wide version of this pipeline -- one traverse and one shade kernel per trip over global
// queues -- was built and measured first: 5.3 ms per headline frame, of which ~3.4 ms was nine
// device-wide waits for each trip's longest ray.  The workgroup-local form below replaced it.)
// ---- Fed launches (round 6).  A persistent launch that is still running when the host is asked for more frames of the same camera takes them itself: the host
// PUBLISHES frames in a block of host-coherent memory (FeedHost) -- seed and sample planes first, then a compare-and-swap on frames_pub -- and a workgroup that finds few
// tiles left LOOKS there (a system-scope load across PCIe, ~1.3 us: profiles/r06_hostfeed_probe.txt), copies what is new into the launch's device mirror (FeedDev) and
// raises frames_known; tiles are frame-major, so the tile counter simply runs on into the new frames.  A workgroup with nothing alive and no tile left CLOSES the feed
// with a compare-and-swap on the same word (bit 31): the host's next compare-and-swap fails and it starts a new launch.  The kernel never waits for the host: what it
// finds published it renders, and when it runs dry it closes and drains like any other launch.  Back-to-back glrtx_render / glrtx_render_frames calls therefore run as ONE
// launch -- one ramp and one drain per burst instead of one per call (window.cpp:121-169's cadence without its per-launch cost; glrtx.hip: feed_append).
constexpr int kFeedMaxFrames = 1024;     // frames one fed launch can take (path ids and sample planes permitting: glrtx.hip)
constexpr int kFeedChunkFrames = 16;     // sample planes are allocated in chunks of this many frames, as the burst grows
constexpr int kFeedChunks = kFeedMaxFrames / kFeedChunkFrames;
constexpr unsigned kFeedClosed = 0x80000000u;
struct FeedHost {  // host-coherent (hipHostMallocMapped | hipHostMallocCoherent); written by the host, except for the closing bit
    unsigned frames_pub;  // bits 0-30: frames published; bit 31: closed by the device
    unsigned pad[31];
    float2 seeds[kFeedMaxFrames];
    float4 *chunks[kFeedChunks];  // sample planes of frames [16 k, 16 k + 16): [frame & 15][sample][owned_rows][pitch_f4]
};
// The device mirror.  Its entries are written while the kernel runs and read through the CU's vector cache like any launch's seeds: a workgroup that sees the count of
// known frames grow invalidates that cache ONCE, before it reads any of the new entries (wg_feed_topup) -- a line fetched for a neighbouring entry earlier may still hold
// what was there before.  (Tried first: every entry in a 128-byte line of its own -- a wave whose paths belong to twenty frames then touched twenty lines for its seeds --
// and reads past the cache by agent-scope loads; both cost a lone launch ~1 %.)
struct FeedDev {
    unsigned frames_known;  // frames copied to this mirror (monotonic; agent-scope atomics); bit 31: the feed is closed and the count final
    unsigned pad[31];
    unsigned long long seeds[kFeedMaxFrames];  // {seed.x, seed.y} as one 64-bit word
    unsigned long long chunks[kFeedChunks];    // device address of the chunk's planes
};
constexpr int kCamFloatsPadded = 36;  // the camera block in LDS: kCamFloats rounded up to float4s
constexpr int kLdsSeeds = 64;     // frames whose seeds a launch keeps in LDS (WfArgs::seeds_in_lds)
constexpr int kWfSetPlanes = 6;  // planes of one set of WfArgs::state (kWfStatePlanes counts both sets)
struct WfArgs {
    // Path state: six float4-wide planes of `ids` entries each, TWICE (one set read, one written per trip), in ONE allocation (one base pointer and one stride
    // in scalar registers instead of twelve pointers).  A path's entry is addressed by WHERE THE PATH STANDS IN ITS WORKGROUP'S PATH QUEUE, not by its pixel:
    // state index = set * 6 * ids + workgroup * block_paths + queue position, plane k at state[k * ids + state index].  The shade phase reads set `cur` at the
    // positions of this trip's queue and writes set `cur ^ 1` at the positions of the next trip's: a wave's 64 loads / stores of a plane are 64 consecutive float4
    // (1 KiB, eight lines) whatever has become of the paths in between -- addressed by pixel they spread over three times as many lines once paths had died --
    // and the allocation is 2 x 6 x 16 B x (workgroups x block_paths) = 0.8 GB whatever the number of frames in flight (by pixel: 3.2 GB at 16 frames, 9.5 GB at 48).
    //   0: {next ray origin (= shadow ray origin), rng.x}      1: {next ray direction, rng.y}
    //   2: {beta, meta}   meta = depth | sample << 8 | flags   3: {L if the pending light sample is accepted (or L), path id = frame * total + tile-order pixel}
    //   4: {L if it is rejected, -}                             5: closest hit of the path's ray {t, tri, u, v}   (written by the traversal lane: the ray
    //      record carries the state index its path will be read at; a parked ray's record is re-addressed when its path moves: wg_shade_phase)
    // (a shadow ray leaves no record in memory: the traversal lane evaluates the light test of :367 itself and sets one bit,
    //  indexed by the owning path's position in the workgroup's path queue, in LDS)
    // (ray origins/directions for the traversal travel in the workgroup's ray queue: 32-byte records
    //  {origin, ray id} {direction, -} in queue order, read with unit stride)
    float4 *state;
    size_t ids;
    // The plane offset is recomputed where it is used (an opaque move keeps the compiler from hoisting seven derived base
    // pointers out of the persistent loop, where they would sit in scalar registers the traversal loop has to spill).
    DEV size_t plane(int k) const { size_t n = ids; asm volatile("" : "+s"(n)); return (size_t)k * n; }
    // (byte offset formed in 32 bits -- the allocation is below 4 GiB -- so that an access can take the scalar plane base plus ONE vector offset register)
    DEV float4 *A(int k, unsigned sidx) const { return reinterpret_cast<float4 *>(reinterpret_cast<char *>(state + plane(k)) + (size_t)(sidx * 16u)); }
    DEV float4 *H(unsigned sidx) const { return A(5, sidx); }
    DEV unsigned set_base(int set, int wg) const { size_t n = ids; asm volatile("" : "+s"(n)); return (unsigned)((size_t)set * kWfSetPlanes * n) + (unsigned)wg * (unsigned)block_paths; }
    int total;        // tile-order ids: tiles8_x * tiles8_y * 64
    int tiles8_x;
    int refill_min;   // refill a traversal wave once this many lanes are idle
    int block_paths;  // paths a workgroup keeps alive (power of two, 256 .. kWgPathsMax)
    int gss_div;      // top-up requests are capped at ceil(tiles left / gss_div); 0 = uncapped
    int suspend_max;  // a wave parks its last path rays at the end of a trip when at most this many lanes still run (0: never; kSuspendMax)
    // Bounds of the persistent loop (pt_render_wgwf: "trip guards").  A workgroup that runs more than trip_limit trips without being given a tile, or two trips in a row
    // in which nothing moved, reports {code, workgroup, trips, live paths} through err -- four words of host-coherent memory the context reads when it folds the launch
    // (GLRTX_EDEVICE) -- and leaves the loop, so that a slip in the queue bookkeeping ends as a failed launch, not as a kernel that never ends.
    int trip_limit;
    unsigned *err;
    // fed launches (FeedHost / FeedDev above; both null otherwise).  n_frames is then the most frames the launch can take; seeds and planes are unused.
    FeedHost *feed_host;
    FeedDev *feed_dev;
    size_t feed_plane_f4;  // float4s per sample plane: owned_rows * pitch_f4
    int feed_margin;       // a workgroup looks at the host's word when fewer than this many tiles are known ahead of the tile counter
    // Frames in flight (glrtx_render_frames): n_frames consecutive frames that differ only in u_seed run in ONE launch.
    // Path ids are frame * total + tile-order pixel id; every finished sample is stored in its own plane
    // (frame * n_samples + sample) and accumulate_planes_kernel adds the planes to the accumulator in frame order, so
    // the sums are formed in exactly the order consecutive launches would form them.  n_frames == 1: seeds are unused; with planes == nullptr
    // the samples are added to the accumulator directly, otherwise they go to planes as well (single-frame launches that overlap:
    // the accumulator is then only touched by the plane-accumulation pass, in launch order).
    const float2 *seeds;  // u_seed of every frame
    int seeds_in_lds;     // 1: the seeds (at most kLdsSeeds frames) are staged into LDS behind the camera block: one gather less per shaded path
    float4 *planes;       // [n_frames * n_samples][owned_rows][pitch_f4] of {min(L, 100), -}
    int n_frames;
    int tiles_per_frame;  // total >> 6
    // id = frame * total + tile-order pixel id (exact stride: no padding of the state arrays to a power of two)
    DEV void split(int id, int &frame, int &pid) const {
        if (n_frames > 1) { frame = (int)((unsigned)id / (unsigned)total); pid = id - frame * total; }
        else { frame = 0; pid = id; }
    }
};
// meta word of the path state: depth in bits 0-7 (it reaches max_depth before the path ends), sample index in bits 8-27.
// The host sends launches beyond these ranges to the persistent megakernel (glrtx_render).
constexpr int kWfStatePlanes = 2 * kWfSetPlanes;  // float4-wide planes of WfArgs::state: two sets of six
constexpr int kWfDepthMax = 255;
constexpr int kWfSampleMax = (1 << 20) - 1;
constexpr unsigned WF_PENDING = 1u << 28;    // a shadow ray of the previous bounce is in flight
constexpr unsigned WF_FINISHING = 1u << 29;  // the path has ended; only that shadow ray is awaited
constexpr unsigned WF_RESOLVED = 1u << 30;   // the pending shadow ray's verdict is known already (bit 31) -- set when a path is deferred (below)
constexpr unsigned WF_ACCEPTED = 1u << 31;
constexpr unsigned WF_INVALID = 0xFFFFFFFFu; // queue entry to skip
// Suspended rays.  When a wave's share of the trip's ray queue is used up, its last few lanes would go on alone until their (long) rays
// are finished -- 12 % of the stepping trips of the headline config ran that way, on 13 lanes of 64 at their start.  Instead, once only
// PATH rays are left in at most kSuspendMax lanes, the wave parks them: the traversal state goes to the workgroup's suspend area (the
// per-lane LDS stacks stay where they are), the path's hit record gets the marker kHitSuspended, and the shade phase defers such a
// path to the next trip -- no shading, no new rays, the verdict of a shadow ray that did finish carried in its state word -- where
// the same lane picks the ray up again next to a full wave of new ones.  Only scheduling changes: every ray runs the same steps.
constexpr int kSuspendMax = 24;
constexpr float kHitSuspended = -1.0f;       // Hit::t of a suspended path ray (a real t exceeds PT_EPS, a miss is PT_INFTY)
constexpr int kSuspendF4 = 4;                // float4s per lane in the suspend area: {o, rid} {d, stop_d} {tHit, tri, u, v} {cur, sp, slot in use, -}
constexpr int kRefillMin = 16;               // refill a traversal wave once this many lanes are idle
#ifdef GLRTX_FAULT_INJECT
constexpr unsigned kFaultPoison = 0xFFFFFFFEu;
#endif

DEV bool wf_pixel(const KernelArgs &a, const WfArgs &w, int id, int &lx, int &lrow) {
    int frame, pid;
    w.split(id, frame, pid);
    const int t = pid >> 6, k = pid & 63;
    lx = (t % w.tiles8_x) * 8 + (k & 7);
    lrow = (t / w.tiles8_x) * 8 + (k >> 3);
    return lx < a.width && lrow < a.owned_rows;
}

// u_seed of the frame a path belongs to
DEV float2 wf_seed(const KernelArgs &a, const WfArgs &w, const float *cam, int id) {
    if (w.n_frames > 1) {
        int frame, pid;
        w.split(id, frame, pid);
        if (w.seeds_in_lds) {  // staged behind the camera block (pt_render_wgwf)
            const float *sd = cam + kCamFloatsPadded + 2 * frame;
            return make_float2(sd[0], sd[1]);
        }
        if (w.feed_dev != nullptr) {  // fed launch: the frame's line of the device mirror
            const nfloat2 v = ((glb_cf2)&w.feed_dev->seeds[0])[frame];
            return make_float2(v.x, v.y);
        }
        const nfloat2 v = ((glb_cf2)w.seeds)[frame];
        return make_float2(v.x, v.y);
    }
    return make_float2(a.seed_x, a.seed_y);
}

// A finished sample: radiance() returns min(L, 100) (:558); main() adds it and counts the sample (:608-609).
// Single frame: read-modify-write of the accumulator.  Frames in flight: the value goes to the sample's plane.
DEV void wf_add_sample(const KernelArgs &a, const WfArgs &w, int id, int lx, int lrow, unsigned sample, float Lx, float Ly, float Lz) {
    if (w.feed_dev != nullptr) {  // fed launch: the frame's chunk of sample planes
        int frame, pid;
        w.split(id, frame, pid);
        typedef __attribute__((address_space(1))) nfloat4 *glb_f4;
        const glb_f4 chunk = (glb_f4)w.feed_dev->chunks[frame / kFeedChunkFrames];  // (a global, not a generic, address)
        const size_t slot = (size_t)(frame % kFeedChunkFrames) * (size_t)a.n_samples + sample;
        nfloat4 v; v.x = fmin_c(Lx, 100.0f); v.y = fmin_c(Ly, 100.0f); v.z = fmin_c(Lz, 100.0f); v.w = 1.0f;
        __builtin_nontemporal_store(v, &chunk[slot * w.feed_plane_f4 + (size_t)lrow * (size_t)a.pitch_f4 + lx]);
    } else if (w.planes != nullptr) {  // frames in flight, or a single frame whose launch overlaps its neighbours (glrtx.hip: launch_wgwf)
        int frame, pid;
        w.split(id, frame, pid);
        const size_t slot = (size_t)frame * (size_t)a.n_samples + sample;
        st_stream(&w.planes[(slot * (size_t)a.owned_rows + (size_t)lrow) * (size_t)a.pitch_f4 + lx],
                  make_float4(fmin_c(Lx, 100.0f), fmin_c(Ly, 100.0f), fmin_c(Lz, 100.0f), 1.0f));
    } else {
        float4 *px = a.accum + (size_t)lrow * a.pitch_f4 + lx;
        float4 acc = *px;
        acc.x = acc.x + fmin_c(Lx, 100.0f);
        acc.y = acc.y + fmin_c(Ly, 100.0f);
        acc.z = acc.z + fmin_c(Lz, 100.0f);
        acc.w = acc.w + 1.0f;
        *px = acc;
    }
}

// Start the pixel's next sample(s): camera ray -> state; returns true if a ray must be traced.
// With u_maxDepth <= 0 a sample is finished as soon as it starts (main() still draws its jitter).
DEV bool wf_start(const KernelArgs &a, const WfArgs &w, const float *cam, int id, int lx, int lrow, Rng &rng, float fcx, float fcy, Path &P, unsigned &sample) {
    for (;;) {
        if ((int)sample >= a.n_samples) return false;
        camera_ray(a, cam, rng, fcx, fcy, P);
        if (a.max_depth > 0) return true;
        wf_add_sample(a, w, id, lx, lrow, sample, 0.0f, 0.0f, 0.0f);  // min(L, 100) of L = 0
        sample++;
    }
}

// Start path `id` (frame | tile-order pixel id): seed its RNG, draw sample 0's camera ray, store the path state.
// Returns true if a ray was queued (false: outside the image, or nothing to trace).
DEV bool wf_generate_one(const KernelArgs &a, const WfArgs &w, const float *cam, int id, unsigned sidx, float4 &ray_o, float4 &ray_d) {
    int lx, lrow;
    bool go = false;
    const float2 sd = wf_seed(a, w, cam, id);
    Rng rng = {0.f, 0.f, sd.x, sd.y};
    Path P;
    unsigned sample = 0;
    if (wf_pixel(a, w, id, lx, lrow)) {  // (ids handed out by the top-up are < n_frames * total by construction)
        const int gy = local_row_to_y(a, lrow);
        const float fcx = (float)lx + 0.5f, fcy = (float)gy + 0.5f;  // gl_FragCoord.xy
        rng.x = fcx / (float)a.width; rng.y = fcy / (float)a.height;  // :567
        go = wf_start(a, w, cam, id, lx, lrow, rng, fcx, fcy, P, sample);
    }
    if (go) {
        st_stream(w.A(0, sidx), make_float4(P.ox, P.oy, P.oz, rng.x));
        st_stream(w.A(1, sidx), make_float4(P.dx, P.dy, P.dz, rng.y));
        st_stream(w.A(2, sidx), make_float4(1.f, 1.f, 1.f, __uint_as_float(sample << 8)));
        st_stream(w.A(3, sidx), make_float4(0.f, 0.f, 0.f, __uint_as_float((unsigned)id)));  // (the path id travels in the state: there is no queue of path ids any more)
        ray_o = make_float4(P.ox, P.oy, P.oz, __uint_as_float(sidx * 2u));
        ray_d = make_float4(P.dx, P.dy, P.dz, 0.f);
    } else st_stream(w.A(3, sidx), make_float4(0.f, 0.f, 0.f, __uint_as_float(WF_INVALID)));  // a position the shade phase skips (a pixel outside the image, nothing to trace)
    return go;
}

// One path of the shade stage: resolve the light sample of the previous bounce, then either close the
// sample (and start the pixel's next one) or run shade_hit() on the new hit.  Outputs which rays to
// queue for the next trip: push_ext = the path's next ray, push_sh = this bounce's shadow ray -- and, when the path goes on (either of them, or requeue), its state
// for the next trip, which the caller stores once it knows the path's next queue position: ray_o = plane 0 {origin, rng.x}, ray_d = plane 1 {direction, rng.y},
// st2 / st3 = planes 2 / 3, st4 = plane 4 when has4.
DEV void wf_shade_path(const KernelArgs &a, const WfArgs &w, const float4 *lds_mats, const float *cam, unsigned &id, unsigned sidx, bool light_accepted, bool &push_ext,
                       bool &push_sh, bool &requeue, float4 &ray_o, float4 &ray_d, float4 &ray_sd, float4 &st2, float4 &st3, float4 &st4, bool &has4, unsigned long long &rays) {
    const float4 s0 = ld_stream(w.A(0, sidx)), s1 = ld_stream(w.A(1, sidx)), s2 = ld_stream(w.A(2, sidx)), s3 = ld_stream(w.A(3, sidx));
    // the hit record is fetched with the state, not behind the meta word the state delivers: one round trip less per round of the shade loop
    // (-0.9 % per frame, profiles/r03_ab_shade_phase.txt; unused when the path has ended).  Plain load and store for the hit records:
    // -1.6 % against the non-temporal forms, profiles/r02_ab_flags.txt
    const float4 hh = *w.H(sidx);
    // the path's id comes with its state (plane 3): until round 6 it was read from a queue of path ids first -- a load and, for every path that goes on, a store per
    // trip on the unit that paces the kernel.  A position the top-up marked invalid is loaded like any other (its planes are whatever was there) and dropped here.
    id = __float_as_uint(s3.w);
    if (id == WF_INVALID) return;
    const float2 sd = wf_seed(a, w, cam, (int)id);
    Rng rng = {s0.w, s1.w, sd.x, sd.y};
    const unsigned meta = __float_as_uint(s2.w);
    unsigned sample = (meta >> 8) & 0xFFFFFu;
    Path P;
    P.ox = s0.x; P.oy = s0.y; P.oz = s0.z; P.dx = s1.x; P.dy = s1.y; P.dz = s1.z;
    P.bx = s2.x; P.by = s2.y; P.bz = s2.z;
    P.depth = (int)(meta & 0xFFu);
    // resolve the light sample of the previous bounce (:367, :539)
    P.Lx = s3.x; P.Ly = s3.y; P.Lz = s3.z;
    if (meta & WF_RESOLVED) light_accepted = (meta & WF_ACCEPTED) != 0u;  // the verdict was carried over a deferral
    const bool rejected = (meta & WF_PENDING) != 0u && !light_accepted;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rejected) {  // the traversal lane's verdict on the shadow ray (nee_accepted), one bit per path
        s4 = ld_stream(w.A(4, sidx));
        P.Lx = s4.x; P.Ly = s4.y; P.Lz = s4.z;
    }
    bool ended = (meta & WF_FINISHING) != 0u;
    Shade sh;
    sh.has_shadow = false;
    if (!ended) {
        if (hh.x == kHitSuspended) {
            // the path's ray is parked in a traversal lane (see kSuspendMax): nothing to shade yet.  The path stays in the queue -- its state moves to its next
            // position as it is -- and the verdict of its shadow ray -- which did finish this trip, and whose bit is indexed by THIS trip's queue position -- moves
            // into the state word.  (The parked ray is told where its path has moved to: wg_shade_phase.)
            unsigned m2 = meta;
            if ((meta & WF_PENDING) && !(meta & WF_RESOLVED)) m2 = meta | WF_RESOLVED | (light_accepted ? WF_ACCEPTED : 0u);
            ray_o = s0; ray_d = s1;
            st2 = make_float4(s2.x, s2.y, s2.z, __uint_as_float(m2));
            st3 = s3;
            st4 = s4;
            has4 = rejected;  // (plane 4 moves with the rest then; an accepted or absent sample never reads it)
            requeue = true;
            return;
        }
        Hit h;
        h.t = hh.x; h.tri = __float_as_int(hh.y); h.u = hh.z; h.v = hh.w;
        if (a.hit_hist != nullptr && h.tri >= 0) atomicAdd(&a.hit_hist[h.tri], 1u);  // (a calibration frame only: glrtx_hit_histogram)
        shade_hit(a, lds_mats, rng, P, h, sh);
        if (sh.untraced) rays += (1ull << 32) + 1ull;  // counted as the reference's intersect() call, not traced (Shade::untraced)
        if (sh.ended && !sh.has_shadow) { P.Lx = sh.Lpx; P.Ly = sh.Lpy; P.Lz = sh.Lpz; }
        ended = sh.ended && !sh.has_shadow;  // with a shadow ray in flight the sample closes next trip
    }
    if (ended) {
        // (the pixel is worked out HERE, where a sample closes, not in front of the shading: two registers less to carry through it)
        int lx, lrow;
        wf_pixel(a, w, (int)id, lx, lrow);
        const int gy = local_row_to_y(a, lrow);
        const float fcx = (float)lx + 0.5f, fcy = (float)gy + 0.5f;
        wf_add_sample(a, w, (int)id, lx, lrow, sample, P.Lx, P.Ly, P.Lz);
        sample++;
        push_ext = wf_start(a, w, cam, (int)id, lx, lrow, rng, fcx, fcy, P, sample);  // the pixel's next sample, if any
        if (push_ext) {
            ray_o = make_float4(P.ox, P.oy, P.oz, rng.x);
            ray_d = make_float4(P.dx, P.dy, P.dz, rng.y);
            st2 = make_float4(1.f, 1.f, 1.f, __uint_as_float(sample << 8));
            st3 = make_float4(0.f, 0.f, 0.f, s3.w);
        }
    } else {
        // shade_hit ran and the path goes on and/or awaits its shadow ray
        push_sh = sh.has_shadow;
        push_ext = !sh.ended;
        const unsigned m2 = (unsigned)P.depth | (sample << 8) | (push_sh ? WF_PENDING : 0u) | (sh.ended ? WF_FINISHING : 0u);
        ray_o = make_float4(P.ox, P.oy, P.oz, rng.x);  // the next ray and the shadow ray leave from the same point
        ray_d = make_float4(P.dx, P.dy, P.dz, rng.y);
        st2 = make_float4(P.bx, P.by, P.bz, __uint_as_float(m2));
        st3 = make_float4(sh.Lpx, sh.Lpy, sh.Lpz, s3.w);
        if (push_sh) { st4 = make_float4(sh.Lfx, sh.Lfy, sh.Lfz, 0.f); has4 = true; }
        ray_sd = make_float4(sh.sdx, sh.sdy, sh.sdz, sh.dist);
    }
}

// ------------------------------------------------------------------------------------------ workgroup-local wavefront
// Variant 2 ("wgwf"): the wavefront formulation with every queue and every barrier LOCAL to a workgroup.
// A workgroup keeps up to block_paths paths alive; each trip it (1) tops its free slots up with new pixels, whole
// 8x8 tiles from the launch's tile counter, (2) traverse phase: lanes pull the queued rays, refilled as they finish,
// __syncthreads, (3) shade phase: the live paths, compacted, __syncthreads -- until nothing is alive and the counter
// is exhausted.  There is no device-wide barrier anywhere: a workgroup waiting for its last long ray idles only
// itself, while the other resident workgroups are at other stages (a device-wide pipeline with one traverse and one
// shade kernel per trip lost ~300 us per trip to that wait, nine times a frame).  One launch covers one frame, or
// several frames in flight (WfArgs).  Path state lives in HBM as float4 SoA.
#ifndef GLRTX_STEPS_PER_TRIP
#define GLRTX_STEPS_PER_TRIP 6
#endif
constexpr int kWgCtlWords = 32;   // control words a workgroup keeps in LDS (pt_render_wgwf: ctl)
constexpr int kWgPathsMax = 4096;  // most paths a workgroup keeps alive (sizes its queues); the host picks block_paths <= this so
                                   // that the launch has that many pixels for every resident workgroup
constexpr size_t kWgSuspendAt = 8 * (size_t)kWgPathsMax + (size_t)kWgPathsMax / 2;  // float4 offset of the suspend area in a workgroup's slice
constexpr size_t kWgQueueF4 = kWgSuspendAt + (size_t)kSuspendF4 * kBlockThreads;   // float4 units per workgroup: ray records + path ids [2] + suspend area
#ifndef GLRTX_PRIO_TRAVERSE
#define GLRTX_PRIO_TRAVERSE 3  // s_setprio of a wave in the traverse phase / in the rest of the trip (shade, top-up)
#define GLRTX_PRIO_SHADE 0
#endif
#ifndef GLRTX_WGWF_WAVES
#define GLRTX_WGWF_WAVES 4
#endif

#ifdef GLRTX_PHASE_STATS
// Diagnostic build only: shader-clock cycles thread 0 of every workgroup spent per phase
// [0] generate, [1] traverse (own work), [2] wait at the barrier after traverse, [3] shade, [4] wait after shade
__device__ unsigned long long g_phase_cycles[8];
// per-trip log of every 64th workgroup: {n_rays, n_paths, traverse+wait cycles, shade+wait cycles}, 64 trips at most;
// entry 0 = {trips, start cycle (low 32 bits), end cycle, blocks taken}
__device__ uint4 g_trip_log[16][64];
#define PH_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PH_ADD(i, t0, t1) do { if (threadIdx.x == 0) atomicAdd(&g_phase_cycles[i], (t1) - (t0)); } while (0)
#else
#define PH_STAMP(var)
#define PH_ADD(i, t0, t1)
#endif

#ifdef GLRTX_RAY_LOG
// Diagnostic build only (-DGLRTX_RAY_LOG, tools/gpu_replay.py): every traverse phase of pt_render_wgwf appends its workgroup's ray queue to a
// global log -- the records as the phase reads them, one (offset, count) pair per trip -- and pt_replay_traverse runs the traverse phase ALONE over
// that log: the same rays in the same grouping, no path state, no shade phase, no top-up.  What it answers: the L2 hit
// rate of the node fetches by themselves, and what a wave-step costs when nothing else streams through the caches.
struct RayLog { float4 *rays; uint2 *trips; unsigned long long cap_rays; unsigned cap_trips; unsigned on; };
__device__ RayLog g_ray_log;
__device__ unsigned long long g_ray_log_n;  // ray records appended (including the ones that did not fit)
__device__ unsigned g_ray_log_trips;        // trips appended
#endif

// Traverse phase of one trip, run by a whole workgroup: lanes pull the workgroup's queued rays and a lane whose ray
// is finished takes the next one once refill_min lanes of its wave are idle.  Path-ray hits go to w.H, shadow-ray verdicts to light_bits.
// The queue holds the rays themselves (32-byte records {origin, ray id} {direction, -}); every wave keeps one chunk of 64
// records in registers -- lane l holds record l -- fetched with unit stride (64 at a time through *ray_head, an LDS
// counter).  A refill therefore touches no memory: the idle lane with rank r takes the record held by lane
// (consumed + r) through a cross-lane read.  (Fetching a ray when a lane fell idle -- queue index, then path state, two
// dependent round trips -- cost 14 % of the phase.  A second chunk fetched ahead of need was kept in round 1; measured
// again in round 2, after the kernel had lost its spills, it bought nothing and its 8 registers were freed; that A/B table was not kept.)
DEV int wgwf_suspend_max();  // WfArgs::suspend_max of the running pt_render_wgwf launch, from its kernarg segment (defined below)

template <bool VINE, int FETCH>
DEV void wg_traverse_phase(const KernelArgs &a, const WfArgs &w, const float4 *root, int *stack, const float4 *rq, int n_rays,
                           unsigned *ray_head, unsigned *light_bits, unsigned long long &rays, float4 *suspend_area) {
    const int lane = threadIdx.x & 63;
    if (VINE) {  // list scan: every ray takes the same number of steps, so waves simply take 64 rays at a time
        for (;;) {
            int base = 0;
            if (lane == 0) base = (int)atomicAdd(ray_head, 64u);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= n_rays) break;
            float4 o = make_float4(0.f, 0.f, 0.f, __uint_as_float(WF_INVALID)), d = o;
            if (base + lane < n_rays) { o = ld_stream(&rq[2 * (size_t)(base + lane)]); d = ld_stream(&rq[2 * (size_t)(base + lane) + 1]); }
            const unsigned r = __float_as_uint(o.w);
            const bool shadow = (r & 1u) != 0u;
            const Hit h = trav_scan<true>(a.sc, o.x, o.y, o.z, d.x, d.y, d.z, r != WF_INVALID, shadow ? shadow_limit(d.w, a.sc.shadow_limited) : PT_INFTY,
                                          shadow ? d.w : -__builtin_inff());
            if (r != WF_INVALID) {
                rays++;
                if (r & 1u) { if (nee_accepted(d.w, h.t, h.tri >= 0)) atomicOr(&light_bits[r >> 6], 1u << ((r >> 1) & 31u)); }
                else st_stream(w.H(r >> 1), make_float4(h.t, __int_as_float(h.tri), h.u, h.v));
            }
        }
        return;
    }
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const float4 none = make_float4(0.f, 0.f, 0.f, __uint_as_float(WF_INVALID));
    float4 cur_o = none, cur_d = none;  // this lane's record of the wave's current chunk
    float cur_ix = 0.f, cur_iy = 0.f, cur_iz = 0.f;  // ... and 1 / direction (:260), computed when the chunk arrives
    int cur_pos = 0, cur_cnt = 0;       // wave-uniform
    // What a ray needs before its first step -- the three IEEE reciprocals and the test of the root's own box -- is done HERE, once per
    // record with all 64 lanes busy, not at refill time in the few lanes that take a ray: a refill is then seven instructions of setup
    // behind its cross-lane reads.  A ray that misses the root box is marked in the sign of d.w (the light sample's distance: never negative).
    auto fetch = [&](float4 &o, float4 &d) -> int {                    // returns the number of records in the chunk (0: queue exhausted)
        int base = 0;
        if (lane == 0) base = (int)atomicAdd(ray_head, 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        int cnt = n_rays - base;
        cnt = cnt < 0 ? 0 : (cnt > 64 ? 64 : cnt);
        if (lane < cnt) {
            o = ld_stream(&rq[2 * (size_t)(base + lane)]);
            d = ld_stream(&rq[2 * (size_t)(base + lane) + 1]);
            cur_ix = frcp(d.x); cur_iy = frcp(d.y); cur_iz = frcp(d.z);
            const bool shadow = (__float_as_uint(o.w) & 1u) != 0u;
            float t0;
            if (a.sc.root_boxed && !box_pass(root[0], root[1], o.x, o.y, o.z, cur_ix, cur_iy, cur_iz, shadow ? shadow_limit(d.w, a.sc.shadow_limited) : PT_INFTY, t0))
                d.w = __uint_as_float(__float_as_uint(d.w) | 0x80000000u);
        }
        return cnt;
    };
    cur_cnt = fetch(cur_o, cur_d);
    bool exhausted = cur_cnt == 0;
    const bool fetched_any = cur_cnt > 0;  // this wave has new rays to run next to the ones it resumes: parking rays again can pay
    bool active = false;
    // A finished ray's hit record is kept in registers and written when the lane is refilled
    // (or at the end of the phase): a store issued inside the stepping loop would sit in vmcnt
    // and make every following node fetch of the whole wave wait for it.
    bool unsaved = false;
    unsigned rid = 0;
    Trav T;
    T.cur = 0; T.sp = 0;
    T.h.t = PT_INFTY; T.h.tri = -1; T.h.u = 0.f; T.h.v = 0.f;
#ifdef GLRTX_STEP_TIMING
    StepTiming step_timing;
#endif
    float4 *const susp = suspend_area + (size_t)kSuspendF4 * threadIdx.x;
    // Rays parked at the end of the previous trip go on in the lanes -- and on the LDS stacks -- they were in.  Whether a lane holds one is
    // read back from its slot (one load per wave and trip; a flag kept in registers across the shade phase costs the kernel its
    // spill-free allocation).
    {
        const float4 s3 = susp[3];
        if (s3.z != 0.0f) {
            const float4 s0 = susp[0], s1 = susp[1], s2 = susp[2];
            susp[3] = make_float4(0.f, 0.f, 0.f, 0.f);
            T.ox = s0.x; T.oy = s0.y; T.oz = s0.z;
            rid = __float_as_uint(w.H(__float_as_uint(s0.w) >> 1)->y);  // the path's state has moved since: the shade phase left the new ray id behind the mark
#ifdef GLRTX_FAULT_INJECT
            const bool fault_dropped = rid == kFaultPoison;  // (see wg_shade_phase)
#endif
            T.dx = s1.x; T.dy = s1.y; T.dz = s1.z; T.stop_d = s1.w;
            T.ix = frcp(T.dx); T.iy = frcp(T.dy); T.iz = frcp(T.dz);  // :260, as at chunk time
            T.h.t = s2.x; T.h.tri = __float_as_int(s2.y); T.h.u = s2.z; T.h.v = s2.w;
            T.cur = __float_as_int(s3.x); T.sp = __float_as_int(s3.y);
#ifdef GLRTX_TRAV_STATS
            T.iters = 0;
#endif
            active = true;
#ifdef GLRTX_FAULT_INJECT
            if (fault_dropped) active = false;
#endif
        }
    }
    auto save_hit = [&]() {
        const unsigned id = rid >> 1;  // path id (path ray) or the path's position in the workgroup's path queue (shadow ray)
        if (rid & 1u) {  // the light test of :367, decided here (T.stop_d is the light sample's distance): one bit for the shade phase
            if (nee_accepted(T.stop_d, T.h.t, T.h.tri >= 0)) atomicOr(&light_bits[id >> 5], 1u << (id & 31u));
        } else *w.H(id) = make_float4(T.h.t, __int_as_float(T.h.tri), T.h.u, T.h.v);  // plain store: scattered 16-byte writes merge in L2 now and then, non-temporal ones never do (-1.2 %)
        unsaved = false;
    };
#ifdef GLRTX_PHASE_STATS
    unsigned long long ph_refill = 0ull, ph_refills = 0ull, ph_step = 0ull;
#endif
    for (;;) {
        unsigned long long idle = __ballot(!active);
        if ((int)__popcll(idle) >= w.refill_min || idle == ~0ull) {
#ifdef GLRTX_PHASE_STATS
            const unsigned long long rf0 = __builtin_amdgcn_s_memtime();
#endif
            while (idle != 0ull && !exhausted) {
                if (cur_pos >= cur_cnt) {  // chunk used up: the next one
                    cur_cnt = cur_cnt == 64 ? fetch(cur_o, cur_d) : 0;
                    cur_pos = 0;
                    if (cur_cnt == 0) { exhausted = true; break; }
                }
                const int n = __popcll(idle);
                const int avail = cur_cnt - cur_pos;
                const int take = n < avail ? n : avail;
                const int rank = __popcll(idle & lt_mask);
                const int src = (cur_pos + rank) & 63;
                const float ox = __shfl(cur_o.x, src), oy = __shfl(cur_o.y, src), oz = __shfl(cur_o.z, src);
                const unsigned new_rid = (unsigned)__shfl((int)__float_as_uint(cur_o.w), src);
                const float dx = __shfl(cur_d.x, src), dy = __shfl(cur_d.y, src), dz = __shfl(cur_d.z, src);
                const float dist_m = __shfl(cur_d.w, src);  // shadow rays: distance of the light sample; sign bit: the root box is missed
                const float ix = __shfl(cur_ix, src), iy = __shfl(cur_iy, src), iz = __shfl(cur_iz, src);
                if (!active && rank < take) {
                    if (unsaved) save_hit();
                    rid = new_rid;
                    if (rid != WF_INVALID) {
                        rays++;
                        const bool shadow = (rid & 1u) != 0u;
                        const float dist = __builtin_fabsf(dist_m);
                        T.ox = ox; T.oy = oy; T.oz = oz; T.dx = dx; T.dy = dy; T.dz = dz; T.ix = ix; T.iy = iy; T.iz = iz;
                        T.h.t = shadow ? shadow_limit(dist, a.sc.shadow_limited) : PT_INFTY; T.h.tri = -1; T.h.u = 0.f; T.h.v = 0.f;
                        T.stop_d = shadow ? dist : -__builtin_inff();
                        T.sp = 0;
                        T.cur = a.sc.root_ref;
#ifdef GLRTX_TRAV_STATS
                        T.iters = 0;
#endif
                        active = (__float_as_uint(dist_m) & 0x80000000u) == 0u;
                        unsaved = !active;  // root box missed: the (miss) record is already final
                    }
                }
                cur_pos += take;
                idle = __ballot(!active);
            }
#ifdef GLRTX_PHASE_STATS
            ph_refill += __builtin_amdgcn_s_memtime() - rf0; ph_refills += 1ull;
#endif
        }
        if (!__any(active)) {
            if (exhausted) break;
            continue;
        }
        // GLRTX_STEPS_PER_TRIP traversal steps per trip through the loop: cuts the refill bookkeeping (ballots, branches) on
        // the latency-critical instruction stream; a lane that finishes on the first step idles for one step
        if (exhausted && fetched_any) {  // (wave-uniform) nothing left to refill with: park the last few path rays
            const unsigned long long ma = __ballot(active);
            const int suspend_max = wgwf_suspend_max();  // (read from the kernarg segment here: not a scalar register held across the loop)
            if (__ballot(active && (rid & 1u) != 0u) == 0ull && (int)__popcll(ma) <= suspend_max) {
                if (active) {
                    susp[0] = make_float4(T.ox, T.oy, T.oz, __uint_as_float(rid));
                    susp[1] = make_float4(T.dx, T.dy, T.dz, T.stop_d);
                    susp[2] = make_float4(T.h.t, __int_as_float(T.h.tri), T.h.u, T.h.v);
                    susp[3] = make_float4(__int_as_float(T.cur), __int_as_float(T.sp), 1.0f, 0.f);  // .z: the slot holds a ray
                    // (the mark's words are formed HERE: hoisted out of the persistent loop as a constant the compiler kept them in four registers it then spilled)
                    float mark = kHitSuspended, zero = 0.f;
                    asm volatile("" : "+v"(mark), "+v"(zero));
                    *w.H(rid >> 1) = make_float4(mark, zero, zero, zero);  // the shade phase defers this path
                    active = false;
                }
                break;
            }
        }
#ifdef GLRTX_PHASE_STATS
        const unsigned long long sb0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef GLRTX_TRAV_STATS
        {
            const unsigned long long ma = __ballot(active);
            if (lane == 0) {
                atomicAdd(&g_trav_trips[0], 1ull); atomicAdd(&g_trav_trips[1], (unsigned long long)__popcll(ma));
                if (exhausted) { atomicAdd(&g_trav_trips[2], 1ull); atomicAdd(&g_trav_trips[3], (unsigned long long)__popcll(ma)); }
            }
        }
#endif
        if (active) {
#if defined(GLRTX_TRAV_STATS) || defined(GLRTX_CXX_STEP)  // diagnostic / experiment builds: the C++ statement of the step
            bool fin = trav_step<true>(a.sc, stack, T);
#pragma unroll
            for (int k = 1; k < GLRTX_STEPS_PER_TRIP; k++)
                if (!fin) fin = trav_step<true>(a.sc, stack, T);
#else
#ifdef GLRTX_STEP_TIMING
            trav_steps_asm<FETCH>(a.sc, stack, T, step_timing);
#else
            trav_steps_asm<FETCH>(a.sc, stack, T);
#endif
            const bool fin = T.cur == REF_FIN;
#endif
            if (fin) {
                active = false;
                unsaved = true;
            }
        }
#ifdef GLRTX_PHASE_STATS
        ph_step += __builtin_amdgcn_s_memtime() - sb0;
#endif
    }
    if (unsaved) save_hit();
#ifdef GLRTX_PHASE_STATS
    // one set of atomics per phase: per-trip atomics on one address (0.4 M per frame) stretched the kernel threefold
    if (threadIdx.x == 0) { atomicAdd(&g_phase_cycles[5], ph_refill); atomicAdd(&g_phase_cycles[6], ph_refills); atomicAdd(&g_phase_cycles[7], ph_step); }
#endif
#ifdef GLRTX_STEP_TIMING
    atomicAdd(&g_step_timing[0], (unsigned long long)step_timing.n);
    atomicAdd(&g_step_timing[1], (unsigned long long)step_timing.t);
    atomicAdd(&g_step_timing[2], (unsigned long long)step_timing.w);
#endif
}

// Shade phase of one trip, run by a whole workgroup: every live path (pq[0..n_paths))
// goes through wf_shade_path(); the rays and paths of the next trip are appended to rq_next / pq_next
// (wave-aggregated, one LDS atomic per wave and queue on *n_rays_next / *n_paths_next).
//
// (Shading the paths in a stable partition by material class -- diffuse / conductor / nothing to do -- was built and measured in
// round 2: bit-identical, 9 % slower per frame; profiles/r02_shade_sort.txt, removed.)
// Shadow rays: the light test's verdict for the path at queue position i is bit i of `light_bits` (LDS), set by the traversal
// lane that finished the path's shadow ray; the shadow ray pushed here carries the position its path will have in pq_next.
// Path state: the path at queue position i is read at state index cur_base + i and -- if it goes on -- written at next_base + (its position in pq_next); the next
// ray's record carries that index (even ray id: where the traversal lane stores the hit), and a ray parked in a traversal lane finds it behind the mark it left.
DEV void wg_shade_phase(const KernelArgs &a, const WfArgs &w, const float4 *lds_mats, const float *cam, const unsigned *light_bits, int n_paths,
                        float4 *rq_next, unsigned *n_rays_next, unsigned *n_paths_next, unsigned cur_base, unsigned next_base, unsigned long long &rays,
                        unsigned *moved) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int j0 = 0; j0 < n_paths; j0 += kBlockThreads) {
        const int i = j0 + (int)threadIdx.x;
        bool push_ext = false, push_sh = false, requeue = false, has4 = false;
        unsigned id = WF_INVALID;
        float4 ro = make_float4(0.f, 0.f, 0.f, 0.f), rd = ro, rsd = ro, st2 = ro, st3 = ro, st4 = ro;
        const bool light_accepted = i < n_paths && ((light_bits[i >> 5] >> (i & 31)) & 1u) != 0u;
        if (i < n_paths)
            wf_shade_path(a, w, lds_mats, cam, id, cur_base + (unsigned)i, light_accepted, push_ext, push_sh, requeue, ro, rd, rsd, st2, st3, st4, has4, rays);
        const unsigned long long me = __ballot(push_ext), ms = __ballot(push_sh), mq = __ballot(requeue), mp = me | ms | mq;
        const unsigned long long mv = __ballot(id != WF_INVALID);
        unsigned br = 0, bp = 0;
        if (lane == 0) {
            if (mv & ~mq) *moved = 1u;  // a path was shaded or closed (not merely carried over behind a parked ray): the trip guard's sign of life
            if (mp) {
                br = atomicAdd(n_rays_next, (unsigned)(__popcll(me) + __popcll(ms)));
                bp = atomicAdd(n_paths_next, (unsigned)__popcll(mp));
            }
        }
        br = __builtin_amdgcn_readfirstlane(br);
        bp = __builtin_amdgcn_readfirstlane(bp);
        const unsigned pos_next = bp + (unsigned)__popcll(mp & lt_mask);  // this path's position in the next trip's path queue
        const unsigned sidx_next = next_base + pos_next;                  // ... and its state index there
        if (push_ext || push_sh || requeue) {
            st_stream(w.A(0, sidx_next), ro);
            st_stream(w.A(1, sidx_next), rd);
            st_stream(w.A(2, sidx_next), st2);
            st_stream(w.A(3, sidx_next), st3);
            if (has4) st_stream(w.A(4, sidx_next), st4);
        }
        // a parked ray's path has moved: the forwarding address goes where the ray left its mark -- the lane reads it when it resumes (wg_traverse_phase), before this
        // set is written again -- and the ray ends the next traverse phase finished or parked again: either way it writes the hit record at the new index
#ifdef GLRTX_FAULT_INJECT
        // Test build only (libglrtx_fault.so, tests/test_gpu_guard.py): the forwarding address is LOST -- the parked ray finds a poison value and is dropped without
        // writing anything (wg_traverse_phase), and the path finds the parked-ray mark at its new position for ever.  Every access stays inside the buffers; what
        // the build shows is that the trip guards of pt_render_wgwf turn such a slip into a failed launch instead of a kernel that never ends.
        if (requeue) {
            *w.H(cur_base + (unsigned)i) = make_float4(kHitSuspended, __uint_as_float(kFaultPoison), 0.f, 0.f);
            *w.H(sidx_next) = make_float4(kHitSuspended, 0.f, 0.f, 0.f);
        }
#else
        if (requeue) *w.H(cur_base + (unsigned)i) = make_float4(kHitSuspended, __uint_as_float(sidx_next * 2u), 0.f, 0.f);
#endif
        if (push_ext) {
            float4 *r = rq_next + 2 * (size_t)(br + __popcll(me & lt_mask));
            r[0] = make_float4(ro.x, ro.y, ro.z, __uint_as_float(sidx_next * 2u));
            r[1] = make_float4(rd.x, rd.y, rd.z, 0.f);
        }
        if (push_sh) {
            float4 *r = rq_next + 2 * (size_t)(br + __popcll(me) + __popcll(ms & lt_mask));
            r[0] = make_float4(ro.x, ro.y, ro.z, __uint_as_float(pos_next * 2u + 1u));  // shadow ray id: queue position, odd
            r[1] = rsd;
        }
    }
}

// The kernel's by-value arguments as they lie in the kernarg segment (same layout rules as a struct).  The shade and top-up
// phases read their launch constants from there, through a pointer the compiler cannot see through, instead of from the
// by-value copies: those would be loaded once at kernel entry and stay live -- in ~100 scalar registers -- across the
// traversal loop.  Read where they are used they are scalar-cache hits with short live ranges.
struct WgwfKernArgs {
    KernelArgs a;
    WfArgs w;
    unsigned *work_counter;
    float4 *wg_queues;
};
DEV const WgwfKernArgs *wgwf_kernargs() {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();  // constant address space
    asm volatile("" : "+s"(p));  // opaque: loads through it stay behind this point
    return (const WgwfKernArgs *)p;
}

DEV int wgwf_suspend_max() { return wgwf_kernargs()->w.suspend_max; }

// Top-up of a FED launch (FeedHost / FeedDev), run by the workgroup's first wave in place of thread 0's part of pt_render_wgwf's top-up: the tiles known so far are
// frames_known x tiles_per_frame; when few are left the wave looks at the host's word, copies the seeds and plane chunks of newly published frames into the device
// mirror (a lane each) and raises frames_known.  Tiles are claimed with ONE atomic add, as in the plain top-up (a compare-and-swap loop that never over-runs the known
// tiles was the first form: a thousand workgroups retrying on one word took a millisecond per top-up).  An add can run past the known tiles; what it claims beyond them
// is not lost -- the tile counter runs on into frames that may yet be published -- but kept by the workgroup as its PENDING claim (ctl[16..17]) and served, before
// anything else is claimed, once the frames it falls into are known.  A workgroup with nothing alive and nothing to serve closes the feed -- its pending claim then lies
// beyond the last frame and is dropped -- or, if the host was quicker, looks again.
DEV void wg_feed_topup(int kWgPaths, unsigned *ctl, int cur, unsigned *work_counter) {
    const WgwfKernArgs *k = wgwf_kernargs();
    FeedHost *fh = k->w.feed_host;
    FeedDev *fd = k->w.feed_dev;
    // Every atomic here is RELAXED: an acquire at agent scope invalidates the CU's vector cache -- once per trip and workgroup that would cost the traverse phase its BVH
    // lines -- and nothing needs one: the counts are read past the caches by the atomics themselves, and what a count announces lies in lines no cache has seen before
    // (FeedDev) or in host memory (read past the caches as well).  "Closed" travels in bit 31 of the same word as the count, so the two are never seen apart.
    const int lane = threadIdx.x;
    const int tpf = k->w.tiles_per_frame, gss_div = k->w.gss_div;
    const int np = (int)ctl[4 + cur];
    const bool may_take = ctl[7] == 0u && ctl[13] == 0u;
    int want = (kWgPaths - np) >> 6;
    unsigned kw = __hip_atomic_load(&fd->frames_known, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int known = (int)(kw & ~kFeedClosed);
    bool closed = (kw & kFeedClosed) != 0u;
    const int done = (int)__hip_atomic_load(work_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // look at the host's word when guided self-scheduling would start to taper the helpings (fewer than feed_margin = gss_div x 64 tiles known ahead): a host that keeps
    // publishing is then always seen in time, and one that has stopped costs a load per trip for the last few trips only
    // Only every 32nd workgroup looks -- and any workgroup that has nothing alive (it is about to close the feed): the others learn of new frames from the device mirror.
    // A load across PCIe is cheap alone (1.3 us) but not from a thousand workgroups at once: with all of them looking every trip a lone launch ran 4 % slower.
    unsigned wg_id = blockIdx.x;  // (opaque: the test is formed here, not kept in a scalar register across the persistent loop)
    asm volatile("" : "+s"(wg_id));
    const bool looker = (wg_id & 31u) == 0u || np == 0;
    if (looker && may_take && want > 0 && !closed && known * tpf - done < k->w.feed_margin) {
        unsigned v = 0u;
        if (lane == 0) v = __hip_atomic_load(&fh->frames_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
        int pub = (int)(v & ~kFeedClosed);
        pub = pub > kFeedMaxFrames ? kFeedMaxFrames : pub;
        if (pub > known) {
            for (int f = known + lane; f < pub; f += 64) {
                const float sx = __hip_atomic_load(&fh->seeds[f].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), sy = __hip_atomic_load(&fh->seeds[f].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&fd->seeds[f], (unsigned long long)__float_as_uint(sx) | ((unsigned long long)__float_as_uint(sy) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (int c = known / kFeedChunkFrames + lane; c <= (pub - 1) / kFeedChunkFrames; c += 64) {
                unsigned long long ptr = __hip_atomic_load(reinterpret_cast<unsigned long long *>(&fh->chunks[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&fd->chunks[c], ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // every lane's copies have reached the L2 before the count that announces them (rare: once per new frame)
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) __hip_atomic_fetch_max(&fd->frames_known, (unsigned)pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            known = pub;
        }
        if ((v & kFeedClosed) != 0u && pub == known) {  // closed by another workgroup, at this count
            if (lane == 0) __hip_atomic_fetch_max(&fd->frames_known, (unsigned)known | kFeedClosed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            closed = true;
        }
    }
    if (known != (int)ctl[18]) {  // (wave-uniform) more frames than this workgroup knew of: their seeds and chunk addresses must not come out of a stale cache line (FeedDev)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (lane == 0) ctl[18] = (unsigned)known;
    }
    if (lane != 0) return;
    const int n_tiles = known * tpf;
    if (want > 0 && gss_div > 0) {  // guided self-scheduling, as in the plain top-up
        const int left = n_tiles - done;
        const int share = left > 0 ? (left + gss_div - 1) / gss_div : 1;
        want = want < share ? want : share;
    }
    int base = 0, got = 0;
    unsigned p_lo = ctl[16], p_hi = ctl[17];  // the pending claim: tiles [p_lo, p_hi) of frames that were not known when they were claimed
    if (may_take) {
        if (p_lo < p_hi) {
            const int v_hi = (int)p_hi < n_tiles ? (int)p_hi : n_tiles;
            if ((int)p_lo < v_hi) { base = (int)p_lo; got = v_hi - (int)p_lo; p_lo += (unsigned)got; }  // (they were claimed when there was room for them, and nothing has come in since)
        } else if (want > 0 && !closed) {
            base = (int)atomicAdd(work_counter, (unsigned)want);
            const int hi = base + want;
            const int v_hi = hi < n_tiles ? hi : (base < n_tiles ? n_tiles : base);
            got = v_hi - base;
            if (hi > v_hi) { p_lo = (unsigned)v_hi; p_hi = (unsigned)hi; }
        }
    }
    unsigned again = 0u;
    if (may_take && got == 0 && np == 0) {
        // nothing alive and nothing to serve: this workgroup would leave.  Close the feed first -- unless it is closed already, or the host has published more in the
        // meantime (the compare-and-swap fails: look again; the loop in pt_render_wgwf comes straight back here)
        if (!closed) {
            unsigned expect = (unsigned)known;
            if (__hip_atomic_compare_exchange_strong(&fh->frames_pub, &expect, (unsigned)known | kFeedClosed, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) ||
                expect == ((unsigned)known | kFeedClosed)) {
                __hip_atomic_fetch_max(&fd->frames_known, (unsigned)known | kFeedClosed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                closed = true;
            } else again = 1u;
        }
    }
    // (bound of the look-again loop, like the trip guards of pt_render_wgwf: a host that publishes a frame every time this workgroup is about to close keeps it here, and
    //  each such turn hands it tiles -- thousands of turns in a row without one mean the bookkeeping has slipped: report and leave)
    const unsigned turns = again ? ctl[15] + 1u : 0u;
    ctl[15] = turns;
    if (turns > 4096u) {
        unsigned *e = k->w.err;
        __hip_atomic_store(e + 1, (unsigned)known, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(e + 2, turns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(e + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(e, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ctl[13] = 1u;
        again = 0u;
    }
    if (closed && (int)p_lo >= n_tiles) { ctl[7] = 1u; p_lo = p_hi = 0u; }  // closed: `known` is final, every tile below it has been claimed by someone, and what this workgroup still holds lies beyond
    ctl[16] = p_lo; ctl[17] = p_hi;
    ctl[0] = (unsigned)base; ctl[6] = (unsigned)got; ctl[14] = again;
    if (np > 0) ctl[12] = got > 0 ? 0u : ctl[12] + 1u;  // (the trip guard counts trips, not looks)
}

template <bool COUNT_RAYS, bool VINE, int FETCH = 0>
__global__ __launch_bounds__(kBlockThreads, GLRTX_WGWF_WAVES) void pt_render_wgwf(const KernelArgs a, const WfArgs w, unsigned *work_counter, float4 *wg_queues) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    // LDS: materials | stack | ctl[16].  The workgroup's ray/path queues live in its private slice of a
    // global buffer (L2-resident, read and written with unit stride).
    float4 *lds_mats = reinterpret_cast<float4 *>(lds_raw);
    const int mat_f4 = a.sc.lds_head_f4;  // materials, then lights
    unsigned char *pl = lds_raw + (size_t)mat_f4 * sizeof(float4);
    int *stack = reinterpret_cast<int *>(pl) + 2 * threadIdx.x;
    pl += (size_t)2 * a.sc.stack_entries * kBlockThreads * sizeof(int);
    unsigned *ctl = reinterpret_cast<unsigned *>(pl);            // 0: first new tile, 1: ray head, 2..3: nRays[2], 4..5: nPaths[2], 6: new tiles, 7: frame exhausted
                                                                 // trip guards -- 10: a path moved this trip, 11: trips in a row in which nothing did, 12: trips since the last tile, 13: abort
                                                                 // fed launches -- 14: nothing to do but the feed is still open: look again, 15: such turns in a row
    // launch constants that only the refill / camera code reads: kept in LDS, not in scalar registers (the kernel arguments alone
    // would occupy ~100 of the 102 SGPRs and spill into VGPR lanes inside the traversal loop)
    float4 *lds_root = reinterpret_cast<float4 *>(pl + kWgCtlWords * sizeof(unsigned));      // {root_lo, root_hi}
    float *lds_cam = reinterpret_cast<float *>(pl + kWgCtlWords * sizeof(unsigned) + 32);     // {c2w, s2c, aperture, focal}
    unsigned *light_bits = reinterpret_cast<unsigned *>(pl + kWgCtlWords * sizeof(unsigned) + 32 + kCamFloatsPadded * 4 + kLdsSeeds * 8);  // kWgPathsMax bits
    if (threadIdx.x < kCamFloats) lds_cam[threadIdx.x] = a.cam[threadIdx.x];
    if (w.seeds_in_lds && (int)threadIdx.x < 2 * w.n_frames) lds_cam[kCamFloatsPadded + threadIdx.x] = reinterpret_cast<const float *>(w.seeds)[threadIdx.x];  // (n_frames <= kLdsSeeds = 64: 128 floats)
    if (threadIdx.x == 64) lds_root[0] = a.sc.root_lo;
    if (threadIdx.x == 65) lds_root[1] = a.sc.root_hi;
    // per-workgroup slice of the queue buffer: ray records float4[2][2 * block_paths][2], then path ids unsigned[2][block_paths]
    float4 *rayQ = wg_queues + (size_t)blockIdx.x * kWgQueueF4;
    if (a.sc.mats_in_lds) stage_mats(a, lds_mats);
    stage_lights(a, lds_mats);

    const int kWgPaths = w.block_paths;
    unsigned long long rays = 0;  // low half: reference rays, high half: those resolved without a traversal
#ifdef GLRTX_PHASE_STATS
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0 && blockIdx.x / 64 < 16)
        g_trip_log[blockIdx.x / 64][0].y = (unsigned)(__builtin_amdgcn_s_memtime() >> 4);
#endif
    if (threadIdx.x == 0) { ctl[2] = 0u; ctl[3] = 0u; ctl[4] = 0u; ctl[5] = 0u; ctl[7] = 0u; ctl[10] = 0u; ctl[11] = 0u; ctl[12] = 0u; ctl[13] = 0u; ctl[14] = 0u; ctl[15] = 0u; ctl[16] = 0u; ctl[17] = 0u; ctl[18] = 0u; }
    if (!VINE) rayQ[kWgSuspendAt + (size_t)kSuspendF4 * threadIdx.x + 3] = make_float4(0.f, 0.f, 0.f, 0.f);  // no lane holds a parked ray yet (wg_traverse_phase)
    int cur = 0;
    __syncthreads();  // materials staged, ctl initialised
    for (;;) {
        // ---- top-up: the workgroup keeps up to kWgPaths paths alive; free slots are refilled with new pixels, whole tiles
        // at a time, from the frame's tile counter (one atomic per workgroup and trip).  Every trip therefore runs on a
        // (nearly) full set of rays, and workgroups finish together when the counter runs out.
        PH_STAMP(pg0);
        unsigned tid_topup = threadIdx.x;  // (an opaque copy, as below: the lane masks of these tests are then formed here, not held in scalar registers across the whole loop)
        asm volatile("" : "+v"(tid_topup));
        if (wgwf_kernargs()->w.feed_dev != nullptr) {
            if (tid_topup < 64u) wg_feed_topup(kWgPaths, ctl, cur, work_counter);  // fed launch: the first wave (it copies what the host has published side by side)
        } else if (tid_topup == 0u) {
            // 8x8-pixel tiles (64 consecutive tile-order ids each), frame-major; formed here from the kernarg segment, not once in front of the persistent
            // loop, for the same reason as gss_div below
            const int n_tiles = wgwf_kernargs()->w.tiles_per_frame * wgwf_kernargs()->w.n_frames;
            int want = (kWgPaths - (int)ctl[4 + cur]) >> 6;
            int base = 0, got = 0;
            // (gss_div is read from the kernarg segment HERE: taken from the by-value argument the compiler hoists the division's reciprocal and sign
            //  words out of the persistent loop, four scalar registers that the list-scan instantiations then spill around their 32 record registers)
            const int gss_div = wgwf_kernargs()->w.gss_div;
            if (want > 0 && ctl[7] == 0u && gss_div > 0) {
                // guided self-scheduling: towards the end of the frame take smaller helpings, so that the last tiles are
                // spread over all workgroups and they finish together (the peek is racy; it only sizes the request)
                const int left = n_tiles - (int)__hip_atomic_load(work_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int share = left > 0 ? (left + gss_div - 1) / gss_div : 1;
                want = want < share ? want : share;
            }
            if (want > 0 && ctl[7] == 0u && ctl[13] == 0u) {
                base = (int)atomicAdd(work_counter, (unsigned)want);
                got = base < n_tiles ? (want < n_tiles - base ? want : n_tiles - base) : 0;
                if (base + want >= n_tiles) ctl[7] = 1u;  // the frame has no more tiles
            }
            ctl[0] = (unsigned)base; ctl[6] = (unsigned)got; ctl[14] = 0u;
            ctl[12] = got > 0 ? 0u : ctl[12] + 1u;
        }
        __syncthreads();
        if (ctl[13] != 0u) break;  // a trip guard fired at the end of the previous trip (below): the launch is reported as failed, whatever is still alive is dropped
        const bool look_again = ctl[14] != 0u;  // (fed launches) nothing alive, no tile, and the feed could not be closed: the host has just published more
        {
            const int got = (int)ctl[6] * 64, tile0 = (int)ctl[0];
            {   // (an opaque copy of the thread index: the compare is then formed here, not kept as a lane mask in two scalar registers across the whole loop)
                unsigned tid = threadIdx.x;
                asm volatile("" : "+v"(tid));
                if (tid < kWgPathsMax / 32) light_bits[tid] = 0u;  // last read by the previous shade phase (a barrier ago)
            }
            const int nr = (int)ctl[2 + cur], np = (int)ctl[4 + cur];
            float4 *rq_w = rayQ + 2 * ((size_t)cur * 2 * kWgPaths + nr);
            const WgwfKernArgs *kt = wgwf_kernargs();
            const unsigned new_base = kt->w.set_base(cur, (int)blockIdx.x) + (unsigned)np;  // state index of the first new path: set `cur`, behind the live ones
            for (int k = threadIdx.x; k < got; k += kBlockThreads) {
                const int id = tile0 * 64 + k;  // tile g = tile0 + (k >> 6) of the frame-major tile order holds ids 64 g .. 64 g + 63
                float4 ro = make_float4(0.f, 0.f, 0.f, __uint_as_float(WF_INVALID)), rd = ro;
                const bool go = wf_generate_one(kt->a, kt->w, lds_cam, id, new_base + (unsigned)k, ro, rd);  // pixels outside the image leave skip markers
                rq_w[2 * k] = ro;  // (plain stores, like every ray record: see st_stream)
                rq_w[2 * k + 1] = rd;
                (void)go;
            }
            __syncthreads();  // everyone has read the counts
            if (tid_topup == 0u) { ctl[2 + cur] = (unsigned)(nr + got); ctl[4 + cur] = (unsigned)(np + got); ctl[1] = 0u; }
        }
        __syncthreads();  // queues of `cur` complete, state stores visible in the workgroup
        PH_STAMP(pg1);
        PH_ADD(0, pg0, pg1);
        const int n_rays = (int)ctl[2 + cur], n_paths = (int)ctl[4 + cur];
        if (n_paths == 0) {
            if (look_again) continue;  // (every thread read the same word behind the same barrier)
            break;  // nothing alive and nothing left to take
        }
        const float4 *rq = rayQ + 2 * ((size_t)cur * 2 * kWgPaths);

#ifdef GLRTX_RAY_LOG
        if (g_ray_log.on) {  // (wave-uniform) append this trip's ray queue to the log
            if (threadIdx.x == 0) {
                const unsigned long long base = atomicAdd(&g_ray_log_n, (unsigned long long)n_rays);
                ctl[9] = 0u;
                if (base + (unsigned long long)n_rays <= g_ray_log.cap_rays) {
                    const unsigned t = atomicAdd(&g_ray_log_trips, 1u);
                    if (t < g_ray_log.cap_trips) { g_ray_log.trips[t] = make_uint2((unsigned)base, (unsigned)n_rays); ctl[8] = (unsigned)base; ctl[9] = 1u; }
                }
            }
            __syncthreads();
            if (ctl[9]) {
                float4 *dst = g_ray_log.rays + 2 * (size_t)ctl[8];
                for (int i = threadIdx.x; i < 2 * n_rays; i += kBlockThreads) dst[i] = rq[i];
            }
            __syncthreads();
        }
#endif
        // ---- traverse phase: lanes pull rays; a lane whose ray is finished takes the next one
        PH_STAMP(pt0);
        // waves in the (memory-latency-bound) traverse phase issue ahead of waves of other workgroups that are shading:
        // their loads get going earlier (measured 1-2 %)
        __builtin_amdgcn_s_setprio(GLRTX_PRIO_TRAVERSE);
        wg_traverse_phase<VINE, FETCH>(a, w, lds_root, stack, rq, n_rays, &ctl[1], light_bits, rays, rayQ + kWgSuspendAt);
        __builtin_amdgcn_s_setprio(GLRTX_PRIO_SHADE);
        PH_STAMP(pt1);
        __syncthreads();  // all hit records of this trip written
        PH_STAMP(pt2);
        PH_ADD(1, pt0, pt1);
        PH_ADD(2, pt1, pt2);

        // ---- shade phase: the live paths; appends go to the other queue pair
        const WgwfKernArgs *ks = wgwf_kernargs();
        wg_shade_phase(ks->a, ks->w, lds_mats, lds_cam, light_bits, n_paths, rayQ + 2 * ((size_t)(cur ^ 1) * 2 * kWgPaths),
                       &ctl[2 + (cur ^ 1)], &ctl[4 + (cur ^ 1)], ks->w.set_base(cur, (int)blockIdx.x), ks->w.set_base(cur ^ 1, (int)blockIdx.x), rays, &ctl[10]);
        PH_STAMP(ps1);
        __syncthreads();  // everyone has read n_rays/n_paths of `cur` and finished appending
        PH_STAMP(ps2);
        PH_ADD(3, pt2, ps1);
        PH_ADD(4, ps1, ps2);
#ifdef GLRTX_PHASE_STATS
        if (threadIdx.x == 0 && (blockIdx.x & 63) == 0 && blockIdx.x / 64 < 16) {
            uint4 *lg = g_trip_log[blockIdx.x / 64];
            const unsigned n = lg[0].x + 1u;
            if (n < 64u) { lg[n] = make_uint4((unsigned)n_rays, (unsigned)n_paths, (unsigned)(pt2 - pt0), (unsigned)(ps2 - pt2)); lg[0].x = n; }
        }
#endif
        unsigned tid_end = threadIdx.x;  // (opaque, like tid_topup)
        asm volatile("" : "+v"(tid_end));
        if (tid_end == 0u) {
            ctl[2 + cur] = 0u; ctl[4 + cur] = 0u;
            // Trip guards: every trip of a sound launch consumes queued rays or shades / closes a path (a path that waits for a parked ray is the only thing that is
            // carried over untouched, and its ray is finished in the first trip that deals no new rays), and a path is alive for at most
            // n_samples x (max_depth + 2) shaded trips plus the trips it spends parked -- the host's trip_limit allows 64 times that without a new tile.
            const bool idle = n_rays == 0 && ctl[10] == 0u;
            ctl[10] = 0u;
            const unsigned idle_trips = idle ? ctl[11] + 1u : 0u;
            ctl[11] = idle_trips;
            const WgwfKernArgs *kg = wgwf_kernargs();
            const unsigned code = idle_trips >= 2u ? 1u : (ctl[12] > (unsigned)kg->w.trip_limit ? 2u : 0u);
            if (code != 0u) {
                unsigned *e = kg->w.err;
                unsigned wg;  // (moved into a vector register HERE: left to the compiler the copy is made in front of the persistent loop and costs the shade phase a register)
                asm volatile("v_mov_b32 %0, %1" : "=v"(wg) : "s"((unsigned)blockIdx.x));
                __hip_atomic_store(e + 1, wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(e + 2, ctl[12], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(e + 3, ctl[4 + (cur ^ 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(e, code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                ctl[13] = 1u;
            }
        }
        cur ^= 1;
    }
#ifdef GLRTX_PHASE_STATS
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0 && blockIdx.x / 64 < 16)
        g_trip_log[blockIdx.x / 64][0].z = (unsigned)(__builtin_amdgcn_s_memtime() >> 4);
#endif
    flush_rays<COUNT_RAYS>(a, rays);
}

#ifdef GLRTX_RAY_LOG
// The traverse phase alone over a recorded log (see RayLog).  Same LDS layout, launch bounds and kernarg prefix as pt_render_wgwf (wgwf_kernargs()
// reads WfArgs::suspend_max from the kernarg segment; the host passes 0: nothing is parked in a replay).  Hit records go where the recorded ray
// ids point (state plane 5), shadow-ray verdicts into the LDS bits, exactly as in the render kernel; nobody reads them.
template <int FETCH>
__global__ __launch_bounds__(kBlockThreads, GLRTX_WGWF_WAVES) void pt_replay_traverse(const KernelArgs a, const WfArgs w, unsigned *work_counter, float4 *wg_queues,
                                                                                      const float4 *log, const uint2 *trips, int n_trips) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int mat_f4 = a.sc.lds_head_f4;  // materials, then lights
    unsigned char *pl = lds_raw + (size_t)mat_f4 * sizeof(float4);
    int *stack = reinterpret_cast<int *>(pl) + 2 * threadIdx.x;
    pl += (size_t)2 * a.sc.stack_entries * kBlockThreads * sizeof(int);
    unsigned *ctl = reinterpret_cast<unsigned *>(pl);
    float4 *lds_root = reinterpret_cast<float4 *>(pl + kWgCtlWords * sizeof(unsigned));
    unsigned *light_bits = reinterpret_cast<unsigned *>(pl + kWgCtlWords * sizeof(unsigned) + 32 + kCamFloatsPadded * 4 + kLdsSeeds * 8);
    if (threadIdx.x == 64) lds_root[0] = a.sc.root_lo;
    if (threadIdx.x == 65) lds_root[1] = a.sc.root_hi;
    float4 *rayQ = wg_queues + (size_t)blockIdx.x * kWgQueueF4;
    rayQ[kWgSuspendAt + (size_t)kSuspendF4 * threadIdx.x + 3] = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned long long rays = 0;
    __syncthreads();
    for (;;) {
        if (threadIdx.x == 0) { ctl[0] = atomicAdd(work_counter, 1u); ctl[1] = 0u; }
        if (threadIdx.x < kWgPathsMax / 32) light_bits[threadIdx.x] = 0u;
        __syncthreads();
        const unsigned t = ctl[0];
        if (t >= (unsigned)n_trips) break;
        const uint2 tr = trips[t];
        __builtin_amdgcn_s_setprio(GLRTX_PRIO_TRAVERSE);
        wg_traverse_phase<false, FETCH>(a, w, lds_root, stack, log + 2 * (size_t)tr.x, (int)tr.y, &ctl[1], light_bits, rays, rayQ + kWgSuspendAt);
        __builtin_amdgcn_s_setprio(GLRTX_PRIO_SHADE);
        __syncthreads();
    }
    flush_rays<true>(a, rays);
}
#endif

// ------------------------------------------------------------------------------------------ frames in flight
// Adds the sample planes of one glrtx_render_frames launch to the accumulator, plane by plane in frame (and sample)
// order: per pixel the same chain of float additions that consecutive single-frame launches perform.
// Bandwidth-bound: 16 B per plane and pixel in, one 16-B read-modify-write of the accumulator.
__global__ __launch_bounds__(256) void accumulate_planes_kernel(float4 *accum, int pitch_f4, int width, int rows, const float4 *planes,
                                                                int n_planes) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= width || y >= rows) return;
    const size_t at = (size_t)y * pitch_f4 + x, plane = (size_t)rows * pitch_f4;
    float4 acc = accum[at];
    for (int k = 0; k < n_planes; k++) {
        const float4 v = planes[(size_t)k * plane + at];
        acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z;
        acc.w = acc.w + 1.0f;
    }
    accum[at] = acc;
}

// In front of a fed launch's render kernel, on its stream: the frames the launch starts with go into the device mirror by ONE wave (left to the render kernel, every
// workgroup would fetch them across PCIe in its first top-up: a thousand times the same reads).
__global__ __launch_bounds__(64) void feed_prefill_kernel(FeedDev *fd, const FeedHost *fh, int n_frames) {
    const int lane = threadIdx.x;
    for (int f = lane; f < n_frames; f += 64) fd->seeds[f] = (unsigned long long)__float_as_uint(fh->seeds[f].x) | ((unsigned long long)__float_as_uint(fh->seeds[f].y) << 32);
    for (int c = lane; c <= (n_frames - 1) / kFeedChunkFrames; c += 64) fd->chunks[c] = (unsigned long long)fh->chunks[c];
    if (lane == 0) fd->frames_known = (unsigned)n_frames;
}

// The same pass behind a FED launch: the frames are however many the launch ended up taking (FeedDev::frames_known, final once the render kernel has ended) and their
// planes lie in chunks of kFeedChunkFrames frames.  Same order of additions: frame by frame, sample by sample.
__global__ __launch_bounds__(256) void accumulate_feed_kernel(float4 *accum, int pitch_f4, int width, int rows, const FeedDev *fd, int n_samples) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= width || y >= rows) return;
    const size_t at = (size_t)y * pitch_f4 + x, plane = (size_t)rows * pitch_f4;
    const int n_frames = (int)(fd->frames_known & ~kFeedClosed);
    float4 acc = accum[at];
    for (int f = 0; f < n_frames; f++) {
        const float4 *chunk = reinterpret_cast<const float4 *>(fd->chunks[f / kFeedChunkFrames]) + (size_t)(f % kFeedChunkFrames) * (size_t)n_samples * plane;
        for (int k = 0; k < n_samples; k++) {
            const float4 v = chunk[(size_t)k * plane + at];
            acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z;
            acc.w = acc.w + 1.0f;
        }
    }
    accum[at] = acc;
}

// ------------------------------------------------------------------------------------------ resolve
// screen.frag:15-25 (rgb / count, clamp, pow(., 1 / u_gamma)) + the RGBA8 colour buffer it is drawn into and
// saveCurrentFrame's glReadPixels read-back with optional vertical flip (window.cpp:297-317, :383-414).
// Byte-exact with the reference's GL implementation (fixtures tests/golden/screen_*.npz, oracle/pt_oracle.c rs_*):
// pow(x, y) = exp2(log2(x) * y) with that implementation's minimax polynomials and fused multiply-adds, IEEE divisions,
// x86 min/max operand order (a NaN quotient clamps to 0), round-to-nearest-even float -> unorm8.  The accumulator texel is
// read exactly (the reference's GL_LINEAR sampler is exact at power-of-two sizes).
// Bandwidth-bound: 16 B in, 4 B out per pixel.
DEV float rs_log2(float x) {
    const uint32_t i = __float_as_uint(x);
    const float ef = (float)((int)((i & 0x7f800000u) >> 23) - 127);
    const float m = __uint_as_float((i & 0x007fffffu) | 0x3f800000u);
    const float t = (m - 1.0f) / (m + 1.0f);
    const float z = t * t, z2 = z * z;
    const float a = __builtin_fmaf(z2, 0x1.a07ab2p-2f, 0x1.27a642p-1f);
    const float b = __builtin_fmaf(z2, 0x1.9d062cp-2f, 0x1.ec6ff2p-1f);
    const float c = __builtin_fmaf(z2, a, 0x1.715476p+1f);
    const float d = __builtin_fmaf(b, z, c);
    return __builtin_fmaf(t, d, ef);  // x in (0, 1] here: the inf / 0 / negative selects of the general routine cannot trigger
}
DEV float rs_exp2(float t) {
    t = (128.0f < t) ? 128.0f : t;
    t = (-0x1.fbfffep+6f > t) ? -0x1.fbfffep+6f : t;
    const float fl = __builtin_floorf(t);
    const float f = t - fl;
    const float scale = __uint_as_float((uint32_t)((int)fl + 127) << 23);
    const float z = f * f;
    const float a = __builtin_fmaf(z, 0x1.ec320ap-10f, 0x1.c95446p-5f);
    const float b = __builtin_fmaf(z, 0x1.26900cp-7f, 0x1.ebd5a8p-3f);
    const float c = __builtin_fmaf(z, a, 0x1.62e4f6p-1f);
    const float d = __builtin_fmaf(z, b, 1.0f);
    return scale * __builtin_fmaf(c, f, d);
}
DEV unsigned char rs_finish(float L, float inv_gamma, float t) {  // t = (m - 1) / (m + 1) of rs_log2, formed by the caller
    float r = 0.0f;
    if (L != 0.0f) {
        const uint32_t i = __float_as_uint(L);
        const float ef = (float)((int)((i & 0x7f800000u) >> 23) - 127);
        const float z = t * t, z2 = z * z;
        const float a = __builtin_fmaf(z2, 0x1.a07ab2p-2f, 0x1.27a642p-1f);
        const float b = __builtin_fmaf(z2, 0x1.9d062cp-2f, 0x1.ec6ff2p-1f);
        const float c = __builtin_fmaf(z2, a, 0x1.715476p+1f);
        const float d = __builtin_fmaf(b, z, c);
        r = rs_exp2(__builtin_fmaf(t, d, ef) * inv_gamma);
    }
    r = (1.0f < r) ? 1.0f : r;
    const int q = (int)__builtin_rintf(r * 255.0f);  // cvtps2dq: round to nearest even
    return (unsigned char)(q < 0 ? 0 : (q > 255 ? 255 : q));
}
DEV float rs_clamp01(float L) {
    L = (L > 0.0f) ? L : 0.0f;  // maxps(L, 0): 0 when L is NaN
    return (L < 1.0f) ? L : 1.0f;  // minps(L, 1)
}
DEV unsigned char rs_channel(float v, float count, float inv_gamma) {  // the plain statement: IEEE quotients as the compiler expands them
    const float L = rs_clamp01(v / count);
    const float m = __uint_as_float((__float_as_uint(L) & 0x007fffffu) | 0x3f800000u);
    return rs_finish(L, inv_gamma, (m - 1.0f) / (m + 1.0f));
}
// The two quotients of a channel -- v / count and (m - 1) / (m + 1) -- in the short form that is PROVED equal to the IEEE quotient (tools/ubench/quotient.hip.h,
// tools/ubench/div_exact.hip: all 2^46 pairs of significands on the device; rcp_newton = RN(1 / b) for every normal b): q = a * r, e = fma(-b, q, a), s = fma(e, r, q),
// valid when s is a normal number, |a| >= 2^-79 and |b| <= 2^40 -- and a numerator that is +-0 over a positive normal divisor is that zero.  The compiler's expansion
// costs 14 vector instructions and two mode switches per quotient, six quotients a pixel: the resolve pass was paced by them, not by memory (round 6).  A wave in which any
// lane falls outside (count = 0: a texel never rendered; a huge or non-finite sum) takes the plain statement for all its lanes.
DEV float rs_quot(float a, float b, float r, bool &outside) {
    const float q = a * r;
    const float s = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
    const bool zero = a == 0.0f;  // (+-0)
    outside = outside || !(zero || (__builtin_amdgcn_class(s, 0x108) && __builtin_fabsf(a) >= 0x1p-79f));
    return zero ? a : s;
}
DEV uchar4 rs_pixel(float4 v, float inv_gamma) {
    bool outside = !(v.w >= 0x1p-126f && v.w <= 0x1p40f);  // count: a positive normal number up to 2^40 (NaN fails)
    const float rc = rcp_newton(v.w);
    const float Lx = rs_clamp01(rs_quot(v.x, v.w, rc, outside)), Ly = rs_clamp01(rs_quot(v.y, v.w, rc, outside)), Lz = rs_clamp01(rs_quot(v.z, v.w, rc, outside));
    const float mx = __uint_as_float((__float_as_uint(Lx) & 0x007fffffu) | 0x3f800000u), my = __uint_as_float((__float_as_uint(Ly) & 0x007fffffu) | 0x3f800000u),
                mz = __uint_as_float((__float_as_uint(Lz) & 0x007fffffu) | 0x3f800000u);
    // (m - 1) / (m + 1): m in [1, 2), the divisor in [2, 3], the numerator 0 or >= 2^-23
    const float tx = rs_quot(mx - 1.0f, mx + 1.0f, rcp_newton(mx + 1.0f), outside), ty = rs_quot(my - 1.0f, my + 1.0f, rcp_newton(my + 1.0f), outside),
                tz = rs_quot(mz - 1.0f, mz + 1.0f, rcp_newton(mz + 1.0f), outside);
    if (__any(outside)) return make_uchar4(rs_channel(v.x, v.w, inv_gamma), rs_channel(v.y, v.w, inv_gamma), rs_channel(v.z, v.w, inv_gamma), 255);
    return make_uchar4(rs_finish(Lx, inv_gamma, tx), rs_finish(Ly, inv_gamma, ty), rs_finish(Lz, inv_gamma, tz), 255);
}
// A wave takes SEG = 64 * PER consecutive pixels of a row -- PER fully coalesced 1-KiB loads issued back to back, then the arithmetic, then PER coalesced 256-byte
// stores.  The pass is paced by its ~250 vector instructions a pixel as much as by memory (round 6, tools/gpu_aux_roofline.py: 12.1 us at 1080p, 39.6 us at 4K per launch
// in a train of launches; a persistent grid that fetches the next segment ahead was slower: 13.8 / 47 us; one and four pixels per lane: 12.8 / 45 and 13.6 / 41).
template <int PER>
__global__ __launch_bounds__(256) void resolve_kernel(const float4 *accum, int pitch_f4, int width, int rows,
                                                      uchar4 *out, int out_pitch_px, float inv_gamma, int flip) {
    constexpr int SEG = 64 * PER;
    const int lane = threadIdx.x & 63;
    const int segs = (width + SEG - 1) / SEG;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int y = task / segs, x0 = (task - y * segs) * SEG + lane;
    if (y >= rows) return;
    const float4 *row = accum + (size_t)y * pitch_f4;
    float4 v[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const int x = x0 + 64 * j;
        v[j] = x < width ? row[x] : make_float4(0.f, 0.f, 0.f, 1.f);  // (a plain load: the accumulator has just been written and sits in the last-level cache; read non-temporally the pass took 15 instead of 12 us)
    }
    const int oy = flip ? rows - 1 - y : y;
    uchar4 *orow = out + (size_t)oy * out_pitch_px;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const int x = x0 + 64 * j;
        const uchar4 px = rs_pixel(v[j], inv_gamma);  // (every lane of the wave: rs_pixel votes)
        if (x < width) orow[x] = px;
    }
}
constexpr int kResolvePer = 2;
inline dim3 resolve_grid(int width, int rows, int per = kResolvePer) { return dim3((unsigned)(((size_t)((width + 64 * per - 1) / (64 * per)) * (size_t)rows + 3) / 4)); }

}  // namespace glrtx
