// scan_asm.hip.h -- the list scan of a "vine" tree (BASELINE config 3: brute force over every triangle, glrt_bvh_build_chain) written by hand for gfx950.
//
// All 64 lanes of a wave are at the SAME list position, so a record (16 dwords: fork box, v0, v1 - v0, v2 - v0, triangle id) comes in through the scalar
// cache into 16 SGPRs and every vector instruction takes one of its operands from there.  The arithmetic is tri_test() / box_pass() of pt_kernel.hip.h
// (reference: intersect(Ray, Triangle) raytrace.frag:226-257, intersectBBox :259-274, the cull of :298), the same IEEE operations in the same order; the
// C++ statement stays in trav_scan() for lists whose forks have boxes of their own, and tests/test_gpu_parity.py compares both with the oracle.
//
// Why by hand (profiles/r04_c3_*, r04_ab_list_scan.txt): as compiled, a record cost a SIMD ~57 vector + ~20 scalar instructions + ~7 branches -- the scan sat at
// the SIMDs' issue limit with more than a third of it spent on exec-mask bookkeeping, and computed the whole triangle test for every record although all 64
// lanes miss most triangles after the first barycentric.  Here the common case -- no lane of the wave passes the u test -- is 30 vector instructions, one scalar
// one and one branch that is not taken; everything behind the u test, the commit of a closer hit, the shadow rays' early stop and the next fork's box test (all
// needed only after a hit) sit out of line.  The box test of a list with ONE common fork box -- min(t1u, tHit) >= t0u, a per-ray constant interval -- changes
// its verdict only when tHit does: it is evaluated in front of the loop and behind hits, not per record.  Four sets of 16 SGPRs: two records are worked on
// while the next two are in flight (scalar loads return out of order, so the only wait is lgkmcnt(0): a load is covered by what is issued between it and that
// wait).  Measured and not kept: six sets of ten dwords, three records ahead (the same time: the scan is not waiting for its records, it issues vector
// instructions); the products and differences as v_pk_mul_f32 / v_pk_add_f32 on register pairs, 23 instructions instead of 30 (exact, and 4 % slower: a packed
// instruction takes the SIMD as long as its two results would).
//
// List layout (glrtx.hip: pack_scene): the n - 1 fork records, never-hit records up to a multiple of four (vine_main), the last leaf's record there (its
// box is infinite: tested with the general box test, as the reference reaches that leaf without a test), three never-hit records behind it.
// Registers: v[GLRTX_VB .. GLRTX_VB+21] scratch (as trav_asm.hip.h), s[34:35] the list position, s[36:99] the four record sets -- all clobbered.
// Precondition (both callers: a path ray, stop_d = -inf; a shadow ray, limit = shadow_limit(stop_d) > stop_d): stop_d - limit < EPS, i.e. a ray is not
// "occluded for certain" before it has hit anything; the early stop is therefore evaluated behind hits only.
#pragma once

#include "trav_asm.hip.h"

#define GLRTX_SCAN_SET_SYMS ".set GLRTX_SA0, 36\n\t.set GLRTX_SA1, 52\n\t.set GLRTX_SB0, 68\n\t.set GLRTX_SB1, 84\n\t"
#define GLRTX_SCAN_SCLOBBERS \
    "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", \
    "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", \
    "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"

// Box test of a fork whose box is the list's common box: min(t1u, tHit) >= t0u with the ray's constant slab interval.  Lanes that fail are finished.
#define GLRTX_SCAN_BOX_UNIFORM \
    "v_min_f32 v[GLRTX_VB+0], %[t1u], %[th]\n\t" \
    "v_cmpx_ge_f32 vcc, v[GLRTX_VB+0], %[t0u]\n\t"
// Box test against the record's own box (box_pass): lo s[S+0..2], hi s[S+4..6].
#define GLRTX_SCAN_BOX_RECORD(S) \
    "v_sub_f32 v[GLRTX_VB+0], s[" S "+0], %[ox]\n\t"   /* (lo - o) * (1 / d) */ \
    "v_sub_f32 v[GLRTX_VB+1], s[" S "+1], %[oy]\n\t" \
    "v_sub_f32 v[GLRTX_VB+2], s[" S "+2], %[oz]\n\t" \
    "v_mul_f32 v[GLRTX_VB+0], v[GLRTX_VB+0], %[ix]\n\t" \
    "v_mul_f32 v[GLRTX_VB+1], v[GLRTX_VB+1], %[iy]\n\t" \
    "v_mul_f32 v[GLRTX_VB+2], v[GLRTX_VB+2], %[iz]\n\t" \
    "v_sub_f32 v[GLRTX_VB+4], s[" S "+4], %[ox]\n\t"   /* (hi - o) * (1 / d) */ \
    "v_sub_f32 v[GLRTX_VB+5], s[" S "+5], %[oy]\n\t" \
    "v_sub_f32 v[GLRTX_VB+6], s[" S "+6], %[oz]\n\t" \
    "v_mul_f32 v[GLRTX_VB+4], v[GLRTX_VB+4], %[ix]\n\t" \
    "v_mul_f32 v[GLRTX_VB+5], v[GLRTX_VB+5], %[iy]\n\t" \
    "v_mul_f32 v[GLRTX_VB+6], v[GLRTX_VB+6], %[iz]\n\t" \
    "v_max_f32 v[GLRTX_VB+16], v[GLRTX_VB+4], v[GLRTX_VB+0]\n\t" \
    "v_min_f32 v[GLRTX_VB+0], v[GLRTX_VB+4], v[GLRTX_VB+0]\n\t" \
    "v_max_f32 v[GLRTX_VB+17], v[GLRTX_VB+5], v[GLRTX_VB+1]\n\t" \
    "v_min_f32 v[GLRTX_VB+1], v[GLRTX_VB+5], v[GLRTX_VB+1]\n\t" \
    "v_max_f32 v[GLRTX_VB+18], v[GLRTX_VB+6], v[GLRTX_VB+2]\n\t" \
    "v_min_f32 v[GLRTX_VB+4], v[GLRTX_VB+6], v[GLRTX_VB+2]\n\t" \
    "v_min3_f32 v[GLRTX_VB+16], v[GLRTX_VB+16], v[GLRTX_VB+17], v[GLRTX_VB+18]\n\t"   /* t1 */ \
    "v_max3_f32 v[GLRTX_VB+2], v[GLRTX_VB+0], v[GLRTX_VB+1], v[GLRTX_VB+4]\n\t"      /* t0 */ \
    "v_min_f32 v[GLRTX_VB+16], v[GLRTX_VB+16], %[th]\n\t" \
    "v_cmpx_ge_f32 vcc, v[GLRTX_VB+16], v[GLRTX_VB+2]\n\t"                          /* min(t1, tHit) >= t0 */

// The triangle of record set S up to the u test; exec = the lanes that passed the box test on entry and on exit (= %[alive]).
// v[VB+1..3] p = d x e2, +4 det, +5 1 / det, +6..8 t = o - v0, +9 U, +10 u, +11 a temporary.
#define GLRTX_SCAN_TRI_HEAD(S, ID) \
    "v_mul_f32 v[GLRTX_VB+1], s[" S "+14], %[dy]\n\t"                 /* p = d x e2 */ \
    "v_mul_f32 v[GLRTX_VB+11], s[" S "+13], %[dz]\n\t" \
    "v_sub_f32 v[GLRTX_VB+1], v[GLRTX_VB+1], v[GLRTX_VB+11]\n\t" \
    "v_mul_f32 v[GLRTX_VB+2], s[" S "+12], %[dz]\n\t" \
    "v_mul_f32 v[GLRTX_VB+11], s[" S "+14], %[dx]\n\t" \
    "v_sub_f32 v[GLRTX_VB+2], v[GLRTX_VB+2], v[GLRTX_VB+11]\n\t" \
    "v_mul_f32 v[GLRTX_VB+3], s[" S "+13], %[dx]\n\t" \
    "v_mul_f32 v[GLRTX_VB+11], s[" S "+12], %[dy]\n\t" \
    "v_sub_f32 v[GLRTX_VB+3], v[GLRTX_VB+3], v[GLRTX_VB+11]\n\t" \
    "v_mul_f32 v[GLRTX_VB+4], s[" S "+11], v[GLRTX_VB+3]\n\t"         /* det = (e1.z pz + e1.y py) + e1.x px */ \
    "v_mul_f32 v[GLRTX_VB+11], s[" S "+10], v[GLRTX_VB+2]\n\t" \
    "v_add_f32 v[GLRTX_VB+4], v[GLRTX_VB+4], v[GLRTX_VB+11]\n\t" \
    "v_mul_f32 v[GLRTX_VB+11], s[" S "+9], v[GLRTX_VB+1]\n\t" \
    "v_add_f32 v[GLRTX_VB+4], v[GLRTX_VB+4], v[GLRTX_VB+11]\n\t" \
    "v_rcp_f32 v[GLRTX_VB+5], v[GLRTX_VB+4]\n\t" \
    "v_subrev_f32 v[GLRTX_VB+6], s[" S "+3], %[ox]\n\t"               /* t = o - v0 */ \
    "v_subrev_f32 v[GLRTX_VB+7], s[" S "+7], %[oy]\n\t" \
    "v_subrev_f32 v[GLRTX_VB+8], s[" S "+8], %[oz]\n\t" \
    "v_fma_f32 v[GLRTX_VB+11], -v[GLRTX_VB+4], v[GLRTX_VB+5], 1.0\n\t"   /* 1 / det: v_rcp + one Newton step (rcp_newton) */ \
    "v_fma_f32 v[GLRTX_VB+5], v[GLRTX_VB+11], v[GLRTX_VB+5], v[GLRTX_VB+5]\n\t" \
    "v_mul_f32 v[GLRTX_VB+9], v[GLRTX_VB+8], v[GLRTX_VB+3]\n\t"       /* U = (tz pz + ty py) + tx px */ \
    "v_mul_f32 v[GLRTX_VB+11], v[GLRTX_VB+7], v[GLRTX_VB+2]\n\t" \
    "v_add_f32 v[GLRTX_VB+9], v[GLRTX_VB+9], v[GLRTX_VB+11]\n\t" \
    "v_mul_f32 v[GLRTX_VB+11], v[GLRTX_VB+6], v[GLRTX_VB+1]\n\t" \
    "v_add_f32 v[GLRTX_VB+9], v[GLRTX_VB+9], v[GLRTX_VB+11]\n\t" \
    "v_mul_f32 v[GLRTX_VB+10], v[GLRTX_VB+9], v[GLRTX_VB+5]\n\t"      /* u */ \
    "v_cmpx_nlt_f32_e64 %[tmp], |v[GLRTX_VB+4]|, %[eps]\n\t"          /* !(-EPS < det && det < EPS) */ \
    "v_cmpx_ngt_f32 vcc, 0, v[GLRTX_VB+10]\n\t"                       /* !(u < 0) */ \
    "v_cmpx_nlt_f32 vcc, 1.0, v[GLRTX_VB+10]\n\t"                     /* !(1 < u) */ \
    "s_cbranch_execnz .Lscan_rest" #ID "_%=\n\t"                       /* some lane may still hit: the rest of the test, out of line */ \
    ".Lscan_back" #ID "_%=:\n\t" \
    "s_mov_b64 exec, %[alive]\n\t"

// Out of line, per record set: the rest of the triangle test with the commit of a closer hit and the shadow rays' early stop.
#define GLRTX_SCAN_TRI_REST(S, ID) \
    ".Lscan_rest" #ID "_%=:\n\t" \
    "v_mul_f32 v[GLRTX_VB+12], s[" S "+11], v[GLRTX_VB+7]\n\t"        /* q = t x e1 */ \
    "v_mul_f32 v[GLRTX_VB+11], s[" S "+10], v[GLRTX_VB+8]\n\t" \
    "v_sub_f32 v[GLRTX_VB+12], v[GLRTX_VB+12], v[GLRTX_VB+11]\n\t" \
    "v_mul_f32 v[GLRTX_VB+13], s[" S "+9], v[GLRTX_VB+8]\n\t" \
    "v_mul_f32 v[GLRTX_VB+11], s[" S "+11], v[GLRTX_VB+6]\n\t" \
    "v_sub_f32 v[GLRTX_VB+13], v[GLRTX_VB+13], v[GLRTX_VB+11]\n\t" \
    "v_mul_f32 v[GLRTX_VB+14], s[" S "+10], v[GLRTX_VB+6]\n\t" \
    "v_mul_f32 v[GLRTX_VB+11], s[" S "+9], v[GLRTX_VB+7]\n\t" \
    "v_sub_f32 v[GLRTX_VB+14], v[GLRTX_VB+14], v[GLRTX_VB+11]\n\t" \
    "v_mul_f32 v[GLRTX_VB+15], %[dz], v[GLRTX_VB+14]\n\t"             /* V = (dz qz + dy qy) + dx qx */ \
    "v_mul_f32 v[GLRTX_VB+11], %[dy], v[GLRTX_VB+13]\n\t" \
    "v_add_f32 v[GLRTX_VB+15], v[GLRTX_VB+15], v[GLRTX_VB+11]\n\t" \
    "v_mul_f32 v[GLRTX_VB+11], %[dx], v[GLRTX_VB+12]\n\t" \
    "v_add_f32 v[GLRTX_VB+15], v[GLRTX_VB+15], v[GLRTX_VB+11]\n\t" \
    "v_mul_f32 v[GLRTX_VB+16], v[GLRTX_VB+15], v[GLRTX_VB+5]\n\t"     /* v */ \
    "v_add_f32 v[GLRTX_VB+11], v[GLRTX_VB+9], v[GLRTX_VB+15]\n\t" \
    "v_mul_f32 v[GLRTX_VB+11], v[GLRTX_VB+5], v[GLRTX_VB+11]\n\t"     /* inv (U + V): u + v > 1 is tested on it */ \
    "v_cmpx_ngt_f32 vcc, 0, v[GLRTX_VB+16]\n\t"                       /* !(v < 0) */ \
    "v_cmpx_nlt_f32 vcc, 1.0, v[GLRTX_VB+11]\n\t"                     /* !(1 < inv (U + V)) */ \
    "s_cbranch_execz .Lscan_back" #ID "_%=\n\t"                        /* nothing was hit: tHit stands, and so does every lane's early-stop verdict */ \
    "v_mul_f32 v[GLRTX_VB+18], s[" S "+14], v[GLRTX_VB+14]\n\t"       /* t = ((e2.z qz + e2.y qy) + e2.x qx) inv */ \
    "v_mul_f32 v[GLRTX_VB+11], s[" S "+13], v[GLRTX_VB+13]\n\t" \
    "v_add_f32 v[GLRTX_VB+18], v[GLRTX_VB+18], v[GLRTX_VB+11]\n\t" \
    "v_mul_f32 v[GLRTX_VB+11], s[" S "+12], v[GLRTX_VB+12]\n\t" \
    "v_add_f32 v[GLRTX_VB+18], v[GLRTX_VB+18], v[GLRTX_VB+11]\n\t" \
    "v_mul_f32 v[GLRTX_VB+18], v[GLRTX_VB+18], v[GLRTX_VB+5]\n\t" \
    "v_cmpx_nge_f32 vcc, %[eps], v[GLRTX_VB+18]\n\t"                  /* !(EPS >= t) */ \
    "v_cmpx_lt_f32 vcc, v[GLRTX_VB+18], %[th]\n\t"                    /* strictly closer: the first one visited wins a tie (:325) */ \
    "v_mov_b32 %[tri], s[" S "+15]\n\t" \
    "v_mov_b32 %[hu], v[GLRTX_VB+10]\n\t" \
    "v_mov_b32 %[hv], v[GLRTX_VB+16]\n\t" \
    "v_mov_b32 %[th], v[GLRTX_VB+18]\n\t" \
    "s_mov_b64 exec, %[alive]\n\t" \
    "v_sub_f32 v[GLRTX_VB+11], %[sd], %[th]\n\t"                      /* shadow ray: a known occluder ends the scan for the lane */ \
    "v_cmpx_nle_f32 vcc, %[eps], v[GLRTX_VB+11]\n\t"                  /* !(stop_d - tHit >= EPS): the ray goes on */ \
    GLRTX_SCAN_BOX_UNIFORM                                             /* ... if the next fork's box is still within reach (tHit has changed) */ \
    "s_mov_b64 %[alive], exec\n\t" \
    "s_branch .Lscan_back" #ID "_%=\n\t"

// A fork of the common box: its test -- min(t1u, tHit) >= t0u -- changes its verdict only when tHit changes, so it is evaluated once in front of the loop and
// behind every hit (GLRTX_SCAN_TRI_REST), not per record; exec = %[alive] = the lanes still scanning on entry and on exit.
#define GLRTX_SCAN_STEP_UNIFORM(S, ID) GLRTX_SCAN_TRI_HEAD(S, ID)
#define GLRTX_SCAN_STEP_RECORD(S, ID) GLRTX_SCAN_BOX_RECORD(S) "s_mov_b64 %[alive], exec\n\t" GLRTX_SCAN_TRI_HEAD(S, ID)

// The scan of a list with one common fork box.  %[grp] = vine_main / 4 groups of four fork records, then the last leaf's record with its own (infinite) box.
#define GLRTX_SCAN_UNIFORM_ASM \
    GLRTX_ASM_SET_VBASE GLRTX_SCAN_SET_SYMS \
    "s_mov_b64 %[entry], exec\n\t" \
    "s_mov_b64 s[34:35], %[ptr]\n\t" \
    "s_load_dwordx16 s[GLRTX_SA0:GLRTX_SA0+15], s[34:35], 0x0\n\t" \
    "s_load_dwordx16 s[GLRTX_SA1:GLRTX_SA1+15], s[34:35], 0x40\n\t" \
    GLRTX_SCAN_BOX_UNIFORM \
    "s_mov_b64 %[alive], exec\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    ".p2align 6\n\t"                                                  /* the loop's head on a 64-byte line, wherever the code in front of it ends (round 5: 1.8 % of config 3 hung on that) */ \
    ".Lscan_loop_%=:\n\t" \
    "s_load_dwordx16 s[GLRTX_SB0:GLRTX_SB0+15], s[34:35], 0x80\n\t" \
    "s_load_dwordx16 s[GLRTX_SB1:GLRTX_SB1+15], s[34:35], 0xc0\n\t" \
    GLRTX_SCAN_STEP_UNIFORM("GLRTX_SA0", 1) \
    GLRTX_SCAN_STEP_UNIFORM("GLRTX_SA1", 2) \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "s_load_dwordx16 s[GLRTX_SA0:GLRTX_SA0+15], s[34:35], 0x100\n\t" \
    "s_load_dwordx16 s[GLRTX_SA1:GLRTX_SA1+15], s[34:35], 0x140\n\t" \
    GLRTX_SCAN_STEP_UNIFORM("GLRTX_SB0", 3) \
    GLRTX_SCAN_STEP_UNIFORM("GLRTX_SB1", 4) \
    "s_add_u32 s34, s34, 0x100\n\t" \
    "s_addc_u32 s35, s35, 0\n\t" \
    "s_sub_u32 %[grp], %[grp], 1\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "s_cbranch_execz .Lscan_end_%=\n\t"                                /* no lane is still scanning */ \
    "s_cmp_lg_u32 %[grp], 0\n\t" \
    "s_cbranch_scc1 .Lscan_loop_%=\n\t" \
    GLRTX_SCAN_STEP_RECORD("GLRTX_SA0", 5)                            /* the last leaf (index vine_main: fetched by the last pass of the loop) */ \
    "s_branch .Lscan_end_%=\n\t" \
    GLRTX_SCAN_TRI_REST("GLRTX_SA0", 1) \
    GLRTX_SCAN_TRI_REST("GLRTX_SA1", 2) \
    GLRTX_SCAN_TRI_REST("GLRTX_SB0", 3) \
    GLRTX_SCAN_TRI_REST("GLRTX_SB1", 4) \
    GLRTX_SCAN_TRI_REST("GLRTX_SA0", 5) \
    ".Lscan_end_%=:\n\t" \
    "s_mov_b64 exec, %[entry]"
