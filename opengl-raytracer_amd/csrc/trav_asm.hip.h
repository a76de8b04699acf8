// trav_asm.hip.h -- the traversal step of the wavefront kernel, written by hand for gfx950.
//
// trav_steps_asm() runs up to GLRTX_STEPS_PER_TRIP steps of the BVH state machine of trav_step() (pt_kernel.hip.h; reference:
// raytrace.frag:276-335 with intersectBBox :259-274 and intersect(Ray, Triangle) :226-257) for the lanes enabled on entry, in ONE asm
// statement.  It performs the same IEEE operations in the same order as trav_step -- the megakernels keep the C++ statement, and
// tests/test_gpu_parity.py::test_all_kernel_variants_bit_identical compares the two on the device.
//
// Why by hand (profiles/r03_ubench_valu.json): on this chip a scalar instruction costs a SIMD MORE issue time than a vector multiply
// (3.2 against 2.0 clk with four waves resident), a taken branch 5.9, and the two kinds overlap only partly.  The compiler's version of
// the step spent ~45 scalar instructions and ~15 branches per wave-step on exec-mask bookkeeping (bools kept as SGPR pairs, merged and
// re-split around every `if`) next to its ~144 vector instructions.  Here:
//   * lane sets are exec masks produced directly by v_cmp / v_cmpx; the triangle test's seven-term predicate is a chain of v_cmpx
//     (each narrows exec, no scalar instruction), with two early exits for waves whose leaf lanes have all failed;
//   * the hit is committed with four moves under the narrowed exec instead of four selects;
//   * the fork arm runs without a branch around it (nearly every step has fork lanes), the leaf arm is skipped when no lane is at a leaf;
//   * a stack entry is {t0, ref}: the fork arm computes t0 of the left child into the register next to its ref, so a push is one
//     ds_write_b64 of a register pair that already exists, and a pop one ds_read_b64;
//   * the two leaves of a fork with two leaf children are chained: the first one's record names the second (pack_scene), and the leaf arm
//     hands a lane on to it without a fork fetch, a push and a pop in between;
//   * leaf (and absent) children carry an INFINITE box in the packed fork record (pack_scene), so "a leaf child is never box-tested"
//     (raytrace.frag:310-331) needs no test of the ref's sign: the slab test passes by itself and yields t0 = -inf.
//   * round 4 -- a PAIR-COOPERATIVE node fetch (GLRTX_TRAV_STEP_ASM_PAIR; the host selects it for large trees).  What paces the traverse phase is the CU's vector-memory pipe, and what that
//     pipe charges for a gather instruction is set by how many different records the lanes of a quad touch: 37 clk per wave-instruction when every lane reads
//     its own record (today's pattern: ~52 lanes in ~25 records), 24.5 when the two lanes of a pair read ONE record, 16 -- the pipe's floor -- when a quad does
//     (tools/ubench/ta.hip, record patterns; profiles/r04_ubench_ta.txt).  So the two lanes of a pair fetch the EVEN lane's 64-byte record between them with
//     two instructions (each lane two of its four 16-byte pieces), then the ODD lane's record with two more -- still four instructions per step, each touching
//     half as many records -- and exchange what the other one needs with one DPP move per dword: 21 selects / moves and 4 address instructions more per step
//     for a third less time in the pipe.  The arms below run unchanged on the assembled record.
// ~147 vector + ~22 scalar + ~8 branch instructions per step (pair fetch; ~122 + ~13 without).
//
// Register use: 24 scratch registers v[GLRTX_VB .. GLRTX_VB+23] with the pair fetch (v96-v119: +16..+22 the partner's pieces in flight, +23 the second
// address), 22 without.  Of those: 22 scratch registers v[GLRTX_VB .. GLRTX_VB+21] (GLRTX_ASM_VBASE, default 96; clobbered), written below relative to the
// assembler symbol GLRTX_VB; in the default build they are v96-v117: v96-v99 A, v100-v103 B, v104-v106 C, v108-v110 D (the 56-byte record; the arms
// compute in place in it), v107 = REF_FIN, v111 an address / u, v112-v117 temporaries.  gfx950 hazards handled by hand (the assembler does not insert wait
// states into inline asm): one independent instruction between v_rcp_f32 and the first use of its result (trans forwarding).
#pragma once

#ifndef GLRTX_STEPS_PER_TRIP
#define GLRTX_STEPS_PER_TRIP 6
#endif
// First of the 22 scratch VGPRs.  96 leaves v0-v95 to the compiler (four waves per SIMD: 128 VGPRs); an experiment build with five
// waves per SIMD (96 VGPRs) moves the block down (-DGLRTX_ASM_VBASE=72 -DGLRTX_WGWF_WAVES=5).
#ifndef GLRTX_ASM_VBASE
#define GLRTX_ASM_VBASE 96
#endif
static_assert(GLRTX_STEPS_PER_TRIP % 2 == 0, "the alternating form of the node fetch unrolls steps in pairs");
#define GLRTX_STEPS_PER_TRIP_HALF_(n) GLRTX_HALF_##n
#define GLRTX_HALF_2 1
#define GLRTX_HALF_4 2
#define GLRTX_HALF_6 3
#define GLRTX_HALF_8 4
#define GLRTX_STEPS_PER_TRIP_HALF_X(n) GLRTX_STEPS_PER_TRIP_HALF_(n)
#define GLRTX_STEPS_PER_TRIP_HALF GLRTX_STEPS_PER_TRIP_HALF_X(GLRTX_STEPS_PER_TRIP)
#define GLRTX_STR_(x) #x
#define GLRTX_STR(x) GLRTX_STR_(x)
#define GLRTX_ASM_SET_VBASE ".set GLRTX_VB, " GLRTX_STR(GLRTX_ASM_VBASE) "\n\t"
// Both forms of the node fetch are compiled, and a third that alternates them step by step (pt_render_wgwf<*, false, FETCH>); the host picks one per scene
// (glrtx.hip: launch_wgwf; GLRTX_PAIR_FETCH=0/1/2 overrides).
#if GLRTX_ASM_VBASE == 96
#define GLRTX_ASM_VCLOBBERS "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117"
#define GLRTX_ASM_VCLOBBERS_PAIR GLRTX_ASM_VCLOBBERS, "v118", "v119"
#elif GLRTX_ASM_VBASE == 72
#define GLRTX_ASM_VCLOBBERS "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93"
#define GLRTX_ASM_VCLOBBERS_PAIR GLRTX_ASM_VCLOBBERS, "v94", "v95"
#else
#error "GLRTX_ASM_VBASE: add the clobber list for this base"
#endif

// One step.  On entry exec = the lanes still running (== %[act]).  Falls through with exec = %[act] = the lanes still running after the
// step, or jumps to 99 (end of block) when none is left.
// Diagnostic build only (-DGLRTX_STEP_TIMING, tools/gpu_steptime.py): per lane, the shader clocks its steps took and the part of them
// spent in the s_waitcnt behind the node fetch.  Uses s90-s95 (clobbered) and three more accumulator operands.
#ifdef GLRTX_STEP_TIMING
#define GLRTX_TS_BEGIN "s_memtime s[90:91]\n\t"
#define GLRTX_TS_WAIT0 "s_memtime s[92:93]\n\t"
#define GLRTX_TS_WAIT1 "s_memtime s[94:95]\n\ts_waitcnt lgkmcnt(0)\n\ts_sub_u32 s92, s94, s92\n\t"
#define GLRTX_TS_END "s_memtime s[94:95]\n\ts_waitcnt lgkmcnt(0)\n\ts_sub_u32 s94, s94, s90\n\tv_add_u32 %[tt], s94, %[tt]\n\tv_add_u32 %[tw], s92, %[tw]\n\tv_add_u32 %[tn], 1, %[tn]\n\t"
#else
#define GLRTX_TS_BEGIN
#define GLRTX_TS_WAIT0
#define GLRTX_TS_WAIT1
#define GLRTX_TS_END
#endif

// (The step's elasticity to one or two more loads per record, or 16 / 32 more vector multiplies, was measured with hooks at this place in round 3 --
// profiles/r03_traverse_bound.txt, r03_ab_elasticity.txt --; the hooks are gone, `git show 56bc2be:opengl-raytracer_amd/csrc/trav_asm.hip.h` has them.)

// The four loads of a record return in order, a gather instruction apart (~17-25 clk each on a busy CU): the slab arithmetic of each
// 16-byte piece starts as soon as that piece is in -- left lo, left hi, right lo, right hi -- instead of behind the last one
// (-1.1 % per frame with two stages; GLRTX_SINGLE_WAIT restores the single s_waitcnt vmcnt(0) for A/B runs).
#if defined(GLRTX_STEP_TIMING)  // (the tool measures the whole wait)
#define GLRTX_SINGLE_WAIT
#endif
#ifndef GLRTX_SINGLE_WAIT
#define GLRTX_W3 "s_waitcnt vmcnt(3)\n\t"
#define GLRTX_W2 "s_waitcnt vmcnt(2)\n\t"
#define GLRTX_W1 "s_waitcnt vmcnt(1)\n\t"
#define GLRTX_W0 "s_waitcnt vmcnt(0)\n\t"
#else
#define GLRTX_W3 "s_waitcnt vmcnt(0)\n\t"
#define GLRTX_W2
#define GLRTX_W1
#define GLRTX_W0
#endif
#define GLRTX_TRAV_STEP_ASM_PAIR \
    GLRTX_TS_BEGIN \
    "s_andn2_b64 %[tmp], %[act], %[odd]\n\t"   /* ---- pair-cooperative fetch: exec = the running lanes and their pair partners */ \
    "s_lshl_b64 %[tmp], %[tmp], 1\n\t" \
    "s_and_b64 %[pop], %[act], %[odd]\n\t" \
    "s_lshr_b64 %[pop], %[pop], 1\n\t" \
    "s_or_b64 %[tmp], %[tmp], %[pop]\n\t" \
    "s_or_b64 %[pop], %[tmp], %[act]\n\t"   /* (%[pop] holds the pair mask until the fork arm assigns it) */ \
    "s_mov_b64 exec, %[pop]\n\t" \
    "v_cndmask_b32_e64 v[GLRTX_VB+4], -1, %[cur], %[act]\n\t"   /* a partner that is not running fetches record -1 for itself -- the never-hit record, present in every scene (a tree of one leaf has no fork 0) -- its cur is REF_FIN: not an address */ \
    "v_cmp_gt_i32_e64 %[leaf], 0, %[cur]\n\t"   /* lanes at a triangle (only ever used under %[act]) */ \
    "s_nop 0\n\t"   /* (a DPP source written by the vector ALU needs two wait states) */ \
    "v_mov_b32_dpp v[GLRTX_VB+15], v[GLRTX_VB+4] quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"   /* the even lane's ref in both lanes of the pair */ \
    "v_mov_b32_dpp v[GLRTX_VB+23], v[GLRTX_VB+4] quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"   /* the odd lane's */ \
    "v_lshl_add_u32 v[GLRTX_VB+15], v[GLRTX_VB+15], 6, %[biase]\n\t"   /* even lanes: pieces 0 and 2 of the even record, odd lanes: 1 and 3 (+16) */ \
    "v_lshl_add_u32 v[GLRTX_VB+23], v[GLRTX_VB+23], 6, %[biaso]\n\t"   /* even lanes: pieces 1 and 3 of the odd record (+16), odd lanes: 0 and 2 */ \
    "global_load_dwordx4 v[GLRTX_VB+0:GLRTX_VB+3], v[GLRTX_VB+15], %[base]\n\t" \
    "global_load_dwordx4 v[GLRTX_VB+16:GLRTX_VB+19], v[GLRTX_VB+23], %[base]\n\t" \
    "global_load_dwordx3 v[GLRTX_VB+8:GLRTX_VB+10], v[GLRTX_VB+15], %[base] offset:32\n\t" \
    "global_load_dwordx3 v[GLRTX_VB+20:GLRTX_VB+22], v[GLRTX_VB+23], %[base] offset:32\n\t" \
    GLRTX_TS_WAIT0 "s_waitcnt vmcnt(2)\n\t" GLRTX_TS_WAIT1 \
    "v_cndmask_b32_e64 v[GLRTX_VB+4], v[GLRTX_VB+16], v[GLRTX_VB+0], %[odd]\n\t"   /* what the partner needs: piece 1 of ITS record */ \
    "v_cndmask_b32_e64 v[GLRTX_VB+5], v[GLRTX_VB+17], v[GLRTX_VB+1], %[odd]\n\t" \
    "v_cndmask_b32_e64 v[GLRTX_VB+6], v[GLRTX_VB+18], v[GLRTX_VB+2], %[odd]\n\t" \
    "v_cndmask_b32_e64 v[GLRTX_VB+7], v[GLRTX_VB+19], v[GLRTX_VB+3], %[odd]\n\t" \
    "v_cndmask_b32_e64 v[GLRTX_VB+0], v[GLRTX_VB+0], v[GLRTX_VB+16], %[odd]\n\t"   /* A = piece 0 of the lane's own record */ \
    "v_cndmask_b32_e64 v[GLRTX_VB+1], v[GLRTX_VB+1], v[GLRTX_VB+17], %[odd]\n\t" \
    "v_cndmask_b32_e64 v[GLRTX_VB+2], v[GLRTX_VB+2], v[GLRTX_VB+18], %[odd]\n\t" \
    "v_cndmask_b32_e64 v[GLRTX_VB+3], v[GLRTX_VB+3], v[GLRTX_VB+19], %[odd]\n\t" \
    "v_mov_b32_dpp v[GLRTX_VB+4], v[GLRTX_VB+4] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   /* B = piece 1, from the partner */ \
    "v_mov_b32_dpp v[GLRTX_VB+5], v[GLRTX_VB+5] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_mov_b32_dpp v[GLRTX_VB+6], v[GLRTX_VB+6] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_mov_b32_dpp v[GLRTX_VB+7], v[GLRTX_VB+7] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
    "s_andn2_b64 exec, %[act], %[leaf]\n\t"   /* ---- fork arm: exec = lanes at a fork (may be none) */ \
    "v_sub_f32 v[GLRTX_VB+0], v[GLRTX_VB+0], %[ox]\n\t"   /* left child: (lo - o) / d, (hi - o) / d */    \
    "v_sub_f32 v[GLRTX_VB+1], v[GLRTX_VB+1], %[oy]\n\t"                                                            \
    "v_sub_f32 v[GLRTX_VB+2], v[GLRTX_VB+2], %[oz]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+0], v[GLRTX_VB+0], %[ix]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+1], v[GLRTX_VB+1], %[iy]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+2], v[GLRTX_VB+2], %[iz]\n\t"                                                            \
    "v_sub_f32 v[GLRTX_VB+4], v[GLRTX_VB+4], %[ox]\n\t"                                                            \
    "v_sub_f32 v[GLRTX_VB+5], v[GLRTX_VB+5], %[oy]\n\t"                                                            \
    "v_sub_f32 v[GLRTX_VB+6], v[GLRTX_VB+6], %[oz]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+4], v[GLRTX_VB+4], %[ix]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+5], v[GLRTX_VB+5], %[iy]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+6], v[GLRTX_VB+6], %[iz]\n\t"                                                            \
    "v_max_f32 v[GLRTX_VB+16], v[GLRTX_VB+4], v[GLRTX_VB+0]\n\t"                                                                                                                    \
    "v_min_f32 v[GLRTX_VB+0], v[GLRTX_VB+4], v[GLRTX_VB+0]\n\t"                                                                                                                     \
    "v_max_f32 v[GLRTX_VB+17], v[GLRTX_VB+5], v[GLRTX_VB+1]\n\t"                                                                                                                    \
    "v_min_f32 v[GLRTX_VB+1], v[GLRTX_VB+5], v[GLRTX_VB+1]\n\t"                                                                                                                     \
    "v_max_f32 v[GLRTX_VB+18], v[GLRTX_VB+6], v[GLRTX_VB+2]\n\t"                                                                                                                    \
    "v_min_f32 v[GLRTX_VB+4], v[GLRTX_VB+6], v[GLRTX_VB+2]\n\t"                                                                                                                    \
    "v_min3_f32 v[GLRTX_VB+16], v[GLRTX_VB+16], v[GLRTX_VB+17], v[GLRTX_VB+18]\n\t"                             /* t1 */                                                                       \
    "v_max3_f32 v[GLRTX_VB+2], v[GLRTX_VB+0], v[GLRTX_VB+1], v[GLRTX_VB+4]\n\t"                                /* t0 of the left child, next to its ref: v[GLRTX_VB+2:GLRTX_VB+3] = {t0, ref} */              \
    "v_min_f32 v[GLRTX_VB+16], v[GLRTX_VB+16], %[th]\n\t"                                                                                                                  \
    "v_cmp_ge_f32_e64 %[bl], v[GLRTX_VB+16], v[GLRTX_VB+2]\n\t"                             /* min(t1, tHit) >= t0 */                                                      \
    "s_mov_b64 exec, %[pop]\n\t"   /* the pairs again */ \
    "s_waitcnt vmcnt(0)\n\t" \
    "v_cndmask_b32_e64 v[GLRTX_VB+12], v[GLRTX_VB+20], v[GLRTX_VB+8], %[odd]\n\t"   /* what the partner needs: piece 3 of its record */ \
    "v_cndmask_b32_e64 v[GLRTX_VB+13], v[GLRTX_VB+21], v[GLRTX_VB+9], %[odd]\n\t" \
    "v_cndmask_b32_e64 v[GLRTX_VB+14], v[GLRTX_VB+22], v[GLRTX_VB+10], %[odd]\n\t" \
    "v_cndmask_b32_e64 v[GLRTX_VB+8], v[GLRTX_VB+8], v[GLRTX_VB+20], %[odd]\n\t"   /* C = piece 2 of the lane's own record */ \
    "v_cndmask_b32_e64 v[GLRTX_VB+9], v[GLRTX_VB+9], v[GLRTX_VB+21], %[odd]\n\t" \
    "v_cndmask_b32_e64 v[GLRTX_VB+10], v[GLRTX_VB+10], v[GLRTX_VB+22], %[odd]\n\t" \
    "v_mov_b32_dpp v[GLRTX_VB+12], v[GLRTX_VB+12] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   /* D = piece 3, from the partner */ \
    "v_mov_b32_dpp v[GLRTX_VB+13], v[GLRTX_VB+13] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_mov_b32_dpp v[GLRTX_VB+14], v[GLRTX_VB+14] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
    "s_andn2_b64 exec, %[act], %[leaf]\n\t"   /* fork lanes again */ \
    "v_sub_f32 v[GLRTX_VB+8], v[GLRTX_VB+8], %[ox]\n\t"   /* right child: lo v[GLRTX_VB+8..10], hi v[GLRTX_VB+12..14] */    \
    "v_sub_f32 v[GLRTX_VB+9], v[GLRTX_VB+9], %[oy]\n\t"                                                            \
    "v_sub_f32 v[GLRTX_VB+10], v[GLRTX_VB+10], %[oz]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+8], v[GLRTX_VB+8], %[ix]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+9], v[GLRTX_VB+9], %[iy]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+10], v[GLRTX_VB+10], %[iz]\n\t"                                                            \
    "v_sub_f32 v[GLRTX_VB+12], v[GLRTX_VB+12], %[ox]\n\t"                                                            \
    "v_sub_f32 v[GLRTX_VB+13], v[GLRTX_VB+13], %[oy]\n\t"                                                            \
    "v_sub_f32 v[GLRTX_VB+14], v[GLRTX_VB+14], %[oz]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+12], v[GLRTX_VB+12], %[ix]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+13], v[GLRTX_VB+13], %[iy]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+14], v[GLRTX_VB+14], %[iz]\n\t"                                                            \
    "v_max_f32 v[GLRTX_VB+16], v[GLRTX_VB+12], v[GLRTX_VB+8]\n\t"                                                                                                                   \
    "v_min_f32 v[GLRTX_VB+8], v[GLRTX_VB+12], v[GLRTX_VB+8]\n\t"                                                                                                                   \
    "v_max_f32 v[GLRTX_VB+17], v[GLRTX_VB+13], v[GLRTX_VB+9]\n\t"                                                                                                                   \
    "v_min_f32 v[GLRTX_VB+9], v[GLRTX_VB+13], v[GLRTX_VB+9]\n\t"                                                                                                                   \
    "v_max_f32 v[GLRTX_VB+18], v[GLRTX_VB+14], v[GLRTX_VB+10]\n\t"                                                                                                                   \
    "v_min_f32 v[GLRTX_VB+10], v[GLRTX_VB+14], v[GLRTX_VB+10]\n\t"                                                                                                                   \
    "v_min3_f32 v[GLRTX_VB+16], v[GLRTX_VB+16], v[GLRTX_VB+17], v[GLRTX_VB+18]\n\t"                                                                                                            \
    "v_max3_f32 v[GLRTX_VB+8], v[GLRTX_VB+8], v[GLRTX_VB+9], v[GLRTX_VB+10]\n\t"                                                                                                            \
    "v_min_f32 v[GLRTX_VB+16], v[GLRTX_VB+16], %[th]\n\t"                                                                                                                  \
    "v_cmp_ge_f32_e64 %[br], v[GLRTX_VB+16], v[GLRTX_VB+8]\n\t" \
    GLRTX_TRAV_STEP_TAIL
#define GLRTX_TRAV_LANE_LOADS \
    "v_lshl_add_u32 v[GLRTX_VB+15], %[cur], 6, %[bias]\n\t"                                                                                                      \
    "global_load_dwordx4 v[GLRTX_VB+0:GLRTX_VB+3], v[GLRTX_VB+15], %[base]\n\t"                                                                                                  \
    "global_load_dwordx4 v[GLRTX_VB+4:GLRTX_VB+7], v[GLRTX_VB+15], %[base] offset:16\n\t"                                                                                      \
    "global_load_dwordx3 v[GLRTX_VB+8:GLRTX_VB+10], v[GLRTX_VB+15], %[base] offset:32\n\t"                                                                                      \
    "global_load_dwordx3 v[GLRTX_VB+12:GLRTX_VB+14], v[GLRTX_VB+15], %[base] offset:48\n\t"
#define GLRTX_TRAV_LANE_CLASSIFY \
    "v_cmp_gt_i32_e64 %[leaf], 0, %[cur]\n\t"                           /* lanes at a triangle */                                                      \
    "s_andn2_b64 exec, exec, %[leaf]\n\t"                               /* ---- fork arm: exec = lanes at a fork (may be none) */
#define GLRTX_TRAV_LANE_FORK \
    GLRTX_TS_WAIT0 GLRTX_W3 GLRTX_TS_WAIT1                                                                                                    \
    "v_sub_f32 v[GLRTX_VB+0], v[GLRTX_VB+0], %[ox]\n\t"   /* left child: (lo - o) / d as soon as the first load is in, (hi - o) / d after the second */    \
    "v_sub_f32 v[GLRTX_VB+1], v[GLRTX_VB+1], %[oy]\n\t"                                                            \
    "v_sub_f32 v[GLRTX_VB+2], v[GLRTX_VB+2], %[oz]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+0], v[GLRTX_VB+0], %[ix]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+1], v[GLRTX_VB+1], %[iy]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+2], v[GLRTX_VB+2], %[iz]\n\t"                                                            \
    GLRTX_W2                                                                                                                        \
    "v_sub_f32 v[GLRTX_VB+4], v[GLRTX_VB+4], %[ox]\n\t"                                                            \
    "v_sub_f32 v[GLRTX_VB+5], v[GLRTX_VB+5], %[oy]\n\t"                                                            \
    "v_sub_f32 v[GLRTX_VB+6], v[GLRTX_VB+6], %[oz]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+4], v[GLRTX_VB+4], %[ix]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+5], v[GLRTX_VB+5], %[iy]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+6], v[GLRTX_VB+6], %[iz]\n\t"                                                            \
    "v_max_f32 v[GLRTX_VB+16], v[GLRTX_VB+4], v[GLRTX_VB+0]\n\t"                                                                                                                    \
    "v_min_f32 v[GLRTX_VB+0], v[GLRTX_VB+4], v[GLRTX_VB+0]\n\t"                                                                                                                     \
    "v_max_f32 v[GLRTX_VB+17], v[GLRTX_VB+5], v[GLRTX_VB+1]\n\t"                                                                                                                    \
    "v_min_f32 v[GLRTX_VB+1], v[GLRTX_VB+5], v[GLRTX_VB+1]\n\t"                                                                                                                     \
    "v_max_f32 v[GLRTX_VB+18], v[GLRTX_VB+6], v[GLRTX_VB+2]\n\t"                                                                                                                    \
    "v_min_f32 v[GLRTX_VB+4], v[GLRTX_VB+6], v[GLRTX_VB+2]\n\t"                                                                                                                    \
    "v_min3_f32 v[GLRTX_VB+16], v[GLRTX_VB+16], v[GLRTX_VB+17], v[GLRTX_VB+18]\n\t"                             /* t1 */                                                                       \
    "v_max3_f32 v[GLRTX_VB+2], v[GLRTX_VB+0], v[GLRTX_VB+1], v[GLRTX_VB+4]\n\t"                                /* t0 of the left child, next to its ref: v[GLRTX_VB+2:GLRTX_VB+3] = {t0, ref} */              \
    "v_min_f32 v[GLRTX_VB+16], v[GLRTX_VB+16], %[th]\n\t"                                                                                                                  \
    "v_cmp_ge_f32_e64 %[bl], v[GLRTX_VB+16], v[GLRTX_VB+2]\n\t"                             /* min(t1, tHit) >= t0 */                                                      \
    GLRTX_W1                                                                                                                        \
    "v_sub_f32 v[GLRTX_VB+8], v[GLRTX_VB+8], %[ox]\n\t"   /* right child: lo v[GLRTX_VB+8..10] (third load), hi v[GLRTX_VB+12..14] (fourth) */    \
    "v_sub_f32 v[GLRTX_VB+9], v[GLRTX_VB+9], %[oy]\n\t"                                                            \
    "v_sub_f32 v[GLRTX_VB+10], v[GLRTX_VB+10], %[oz]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+8], v[GLRTX_VB+8], %[ix]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+9], v[GLRTX_VB+9], %[iy]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+10], v[GLRTX_VB+10], %[iz]\n\t"                                                            \
    GLRTX_W0                                                                                                                        \
    "v_sub_f32 v[GLRTX_VB+12], v[GLRTX_VB+12], %[ox]\n\t"                                                            \
    "v_sub_f32 v[GLRTX_VB+13], v[GLRTX_VB+13], %[oy]\n\t"                                                            \
    "v_sub_f32 v[GLRTX_VB+14], v[GLRTX_VB+14], %[oz]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+12], v[GLRTX_VB+12], %[ix]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+13], v[GLRTX_VB+13], %[iy]\n\t"                                                            \
    "v_mul_f32 v[GLRTX_VB+14], v[GLRTX_VB+14], %[iz]\n\t"                                                            \
    "v_max_f32 v[GLRTX_VB+16], v[GLRTX_VB+12], v[GLRTX_VB+8]\n\t"                                                                                                                   \
    "v_min_f32 v[GLRTX_VB+8], v[GLRTX_VB+12], v[GLRTX_VB+8]\n\t"                                                                                                                   \
    "v_max_f32 v[GLRTX_VB+17], v[GLRTX_VB+13], v[GLRTX_VB+9]\n\t"                                                                                                                   \
    "v_min_f32 v[GLRTX_VB+9], v[GLRTX_VB+13], v[GLRTX_VB+9]\n\t"                                                                                                                   \
    "v_max_f32 v[GLRTX_VB+18], v[GLRTX_VB+14], v[GLRTX_VB+10]\n\t"                                                                                                                   \
    "v_min_f32 v[GLRTX_VB+10], v[GLRTX_VB+14], v[GLRTX_VB+10]\n\t"                                                                                                                   \
    "v_min3_f32 v[GLRTX_VB+16], v[GLRTX_VB+16], v[GLRTX_VB+17], v[GLRTX_VB+18]\n\t"                                                                                                            \
    "v_max3_f32 v[GLRTX_VB+8], v[GLRTX_VB+8], v[GLRTX_VB+9], v[GLRTX_VB+10]\n\t"                                                                                                            \
    "v_min_f32 v[GLRTX_VB+16], v[GLRTX_VB+16], %[th]\n\t"                                                                                                                  \
    "v_cmp_ge_f32_e64 %[br], v[GLRTX_VB+16], v[GLRTX_VB+8]\n\t"
#define GLRTX_TRAV_STEP_ASM_LANE GLRTX_TS_BEGIN GLRTX_TRAV_LANE_LOADS GLRTX_TRAV_LANE_CLASSIFY GLRTX_TRAV_LANE_FORK GLRTX_TRAV_STEP_TAIL
// (A step without the leaf arm in every other step -- the leaf arm on twice the lanes -- was built in round 5: bit-identical, +6.25 %, profiles/r05_lane_util.txt; removed.)
#define GLRTX_TRAV_STEP_TAIL \
    "v_cndmask_b32_e64 %[cur], v[GLRTX_VB+3], v[GLRTX_VB+7], %[br]\n\t"                    /* go on with the right child if it passed, else with the left */              \
    "s_or_b64 %[tmp], %[bl], %[br]\n\t"                                                                                                                \
    "s_andn2_b64 %[pop], exec, %[tmp]\n\t"                              /* fork lanes with neither child: pop */                                       \
    "s_and_b64 exec, %[bl], %[br]\n\t"                                  /* both passed: the left one waits on the stack */                             \
    "v_lshl_add_u32 v[GLRTX_VB+15], %[sp], 11, %[stk]\n\t"                                                                                                       \
    "ds_write_b64 v[GLRTX_VB+15], v[GLRTX_VB+2:GLRTX_VB+3]\n\t"                                                                                                                  \
    "v_add_u32 %[sp], 1, %[sp]\n\t"                                                                                                                    \
    "s_and_b64 exec, %[act], %[leaf]\n\t"                               /* ---- leaf arm: A = {v0, material} v[GLRTX_VB+0]..99, B = v1 - v0 v[GLRTX_VB+4]..102, C = v2 - v0 v[GLRTX_VB+8]..106 */ \
    "s_cbranch_scc0 21f\n\t"                                                                                                                           \
    "v_not_b32 v[GLRTX_VB+3], %[cur]\n\t"                                         /* triangle index */                                                           \
    "v_mov_b32 %[cur], v[GLRTX_VB+7]\n\t"                                       /* the triangle chained behind this one (the other leaf of a leaf pair), or REF_FIN */            \
    "v_mul_f32 v[GLRTX_VB+16], %[dy], v[GLRTX_VB+10]\n\t"                                   /* p = d x e2 */                                                               \
    "v_mul_f32 v[GLRTX_VB+19], %[dz], v[GLRTX_VB+9]\n\t"                                                                                                                  \
    "v_sub_f32 v[GLRTX_VB+16], v[GLRTX_VB+16], v[GLRTX_VB+19]\n\t"                                                                                                                   \
    "v_mul_f32 v[GLRTX_VB+17], %[dz], v[GLRTX_VB+8]\n\t"                                                                                                                  \
    "v_mul_f32 v[GLRTX_VB+19], %[dx], v[GLRTX_VB+10]\n\t"                                                                                                                  \
    "v_sub_f32 v[GLRTX_VB+17], v[GLRTX_VB+17], v[GLRTX_VB+19]\n\t"                                                                                                                   \
    "v_mul_f32 v[GLRTX_VB+18], %[dx], v[GLRTX_VB+9]\n\t"                                                                                                                  \
    "v_mul_f32 v[GLRTX_VB+19], %[dy], v[GLRTX_VB+8]\n\t"                                                                                                                  \
    "v_sub_f32 v[GLRTX_VB+18], v[GLRTX_VB+18], v[GLRTX_VB+19]\n\t"                                                                                                                   \
    "v_mul_f32 v[GLRTX_VB+20], v[GLRTX_VB+6], v[GLRTX_VB+18]\n\t"                                    /* det = (e1.z pz + e1.y py) + e1.x px */                                      \
    "v_mul_f32 v[GLRTX_VB+19], v[GLRTX_VB+5], v[GLRTX_VB+17]\n\t"                                                                                                                   \
    "v_add_f32 v[GLRTX_VB+20], v[GLRTX_VB+20], v[GLRTX_VB+19]\n\t"                                                                                                                   \
    "v_mul_f32 v[GLRTX_VB+19], v[GLRTX_VB+4], v[GLRTX_VB+16]\n\t"                                                                                                                   \
    "v_add_f32 v[GLRTX_VB+20], v[GLRTX_VB+20], v[GLRTX_VB+19]\n\t"                                                                                                                   \
    "v_rcp_f32 v[GLRTX_VB+21], v[GLRTX_VB+20]\n\t"                                                                                                                         \
    "v_sub_f32 v[GLRTX_VB+0], %[ox], v[GLRTX_VB+0]\n\t"                                     /* t = o - v0, in place */                                                     \
    "v_sub_f32 v[GLRTX_VB+1], %[oy], v[GLRTX_VB+1]\n\t"                                                                                                                    \
    "v_sub_f32 v[GLRTX_VB+2], %[oz], v[GLRTX_VB+2]\n\t"                                                                                                                    \
    "v_fma_f32 v[GLRTX_VB+19], -v[GLRTX_VB+20], v[GLRTX_VB+21], 1.0\n\t"                              /* 1 / det: v_rcp + one Newton step (rcp_newton: the IEEE quotient of every normal det, denormals being flushed) */                             \
    "v_fma_f32 v[GLRTX_VB+21], v[GLRTX_VB+19], v[GLRTX_VB+21], v[GLRTX_VB+21]\n\t"                                                                                                             \
    "v_mul_f32 v[GLRTX_VB+18], v[GLRTX_VB+2], v[GLRTX_VB+18]\n\t"                                     /* U = (tz pz + ty py) + tx px */                                              \
    "v_mul_f32 v[GLRTX_VB+19], v[GLRTX_VB+1], v[GLRTX_VB+17]\n\t"                                                                                                                    \
    "v_add_f32 v[GLRTX_VB+18], v[GLRTX_VB+18], v[GLRTX_VB+19]\n\t"                                                                                                                   \
    "v_mul_f32 v[GLRTX_VB+19], v[GLRTX_VB+0], v[GLRTX_VB+16]\n\t"                                                                                                                    \
    "v_add_f32 v[GLRTX_VB+18], v[GLRTX_VB+18], v[GLRTX_VB+19]\n\t"                                                                                                                   \
    "v_mul_f32 v[GLRTX_VB+15], v[GLRTX_VB+18], v[GLRTX_VB+21]\n\t"                                    /* u */                                                                        \
    "v_cmpx_nlt_f32_e64 %[tmp], |v[GLRTX_VB+20]|, %[eps]\n\t"                     /* !(-EPS < det && det < EPS) */                                               \
    "v_cmpx_ngt_f32 vcc, 0, v[GLRTX_VB+15]\n\t"                                   /* !(u < 0) */                                                                 \
    "v_cmpx_nlt_f32 vcc, 1.0, v[GLRTX_VB+15]\n\t"                                 /* !(1 < u) */                                                                 \
    "s_cbranch_execz 20f\n\t"                                                                                                                          \
    "v_mul_f32 v[GLRTX_VB+12], v[GLRTX_VB+1], v[GLRTX_VB+6]\n\t"                                     /* q = t x e1 */                                                               \
    "v_mul_f32 v[GLRTX_VB+19], v[GLRTX_VB+2], v[GLRTX_VB+5]\n\t"                                                                                                                    \
    "v_sub_f32 v[GLRTX_VB+12], v[GLRTX_VB+12], v[GLRTX_VB+19]\n\t"                                                                                                                   \
    "v_mul_f32 v[GLRTX_VB+13], v[GLRTX_VB+2], v[GLRTX_VB+4]\n\t"                                                                                                                    \
    "v_mul_f32 v[GLRTX_VB+19], v[GLRTX_VB+0], v[GLRTX_VB+6]\n\t"                                                                                                                    \
    "v_sub_f32 v[GLRTX_VB+13], v[GLRTX_VB+13], v[GLRTX_VB+19]\n\t"                                                                                                                   \
    "v_mul_f32 v[GLRTX_VB+14], v[GLRTX_VB+0], v[GLRTX_VB+5]\n\t"                                                                                                                    \
    "v_mul_f32 v[GLRTX_VB+19], v[GLRTX_VB+1], v[GLRTX_VB+4]\n\t"                                                                                                                    \
    "v_sub_f32 v[GLRTX_VB+14], v[GLRTX_VB+14], v[GLRTX_VB+19]\n\t"                                                                                                                   \
    "v_mul_f32 v[GLRTX_VB+7], %[dz], v[GLRTX_VB+14]\n\t"                                   /* V = (dz qz + dy qy) + dx qx */                                              \
    "v_mul_f32 v[GLRTX_VB+19], %[dy], v[GLRTX_VB+13]\n\t"                                                                                                                  \
    "v_add_f32 v[GLRTX_VB+7], v[GLRTX_VB+7], v[GLRTX_VB+19]\n\t"                                                                                                                   \
    "v_mul_f32 v[GLRTX_VB+19], %[dx], v[GLRTX_VB+12]\n\t"                                                                                                                  \
    "v_add_f32 v[GLRTX_VB+7], v[GLRTX_VB+7], v[GLRTX_VB+19]\n\t"                                                                                                                   \
    "v_mul_f32 v[GLRTX_VB+16], v[GLRTX_VB+7], v[GLRTX_VB+21]\n\t"                                    /* v */                                                                        \
    "v_add_f32 v[GLRTX_VB+19], v[GLRTX_VB+18], v[GLRTX_VB+7]\n\t"                                                                                                                   \
    "v_mul_f32 v[GLRTX_VB+19], v[GLRTX_VB+21], v[GLRTX_VB+19]\n\t"                                    /* inv (U + V): u + v > 1 is tested on it */                                   \
    "v_cmpx_ngt_f32 vcc, 0, v[GLRTX_VB+16]\n\t"                                   /* !(v < 0) */                                                                 \
    "v_cmpx_nlt_f32 vcc, 1.0, v[GLRTX_VB+19]\n\t"                                 /* !(1 < inv (U + V)) */                                                       \
    "s_cbranch_execz 20f\n\t"                                                                                                                          \
    "v_mul_f32 v[GLRTX_VB+19], v[GLRTX_VB+10], v[GLRTX_VB+14]\n\t"                                    /* t = ((e2.z qz + e2.y qy) + e2.x qx) inv */                                  \
    "v_mul_f32 v[GLRTX_VB+7], v[GLRTX_VB+9], v[GLRTX_VB+13]\n\t"                                                                                                                   \
    "v_add_f32 v[GLRTX_VB+19], v[GLRTX_VB+19], v[GLRTX_VB+7]\n\t"                                                                                                                   \
    "v_mul_f32 v[GLRTX_VB+7], v[GLRTX_VB+8], v[GLRTX_VB+12]\n\t"                                                                                                                   \
    "v_add_f32 v[GLRTX_VB+19], v[GLRTX_VB+19], v[GLRTX_VB+7]\n\t"                                                                                                                   \
    "v_mul_f32 v[GLRTX_VB+19], v[GLRTX_VB+19], v[GLRTX_VB+21]\n\t"                                                                                                                   \
    "v_cmpx_nge_f32 vcc, %[eps], v[GLRTX_VB+19]\n\t"                              /* !(EPS >= t) */                                                              \
    "v_cmpx_lt_f32 vcc, v[GLRTX_VB+19], %[th]\n\t"                                /* strictly closer: the first one visited wins a tie (:325) */                 \
    "v_mov_b32 %[tri], v[GLRTX_VB+3]\n\t"                                                                                                                        \
    "v_mov_b32 %[hu], v[GLRTX_VB+15]\n\t"                                                                                                                        \
    "v_mov_b32 %[hv], v[GLRTX_VB+16]\n\t"                                                                                                                        \
    "v_mov_b32 %[th], v[GLRTX_VB+19]\n\t"                                                                                                                        \
    "20:\n\t"                                                                                                                                          \
    "s_and_b64 exec, %[act], %[leaf]\n\t"                               /* every leaf lane again */                                                    \
    "v_sub_f32 v[GLRTX_VB+16], %[sd], %[th]\n\t"                                  /* shadow ray: a known occluder ends the traversal */                          \
    "v_cmp_nle_f32 vcc, %[eps], v[GLRTX_VB+16]\n\t"                              /* !(stop_d - tHit >= EPS): the ray goes on */                                 \
    "v_cndmask_b32 %[cur], v[GLRTX_VB+11], %[cur], vcc\n\t"                      /* ... with the chained triangle if there is one; a stopped ray is finished */ \
    "v_cmp_eq_u32_e64 %[tmp], %[cur], v[GLRTX_VB+11]\n\t"                                                                                                       \
    "s_and_b64 vcc, vcc, %[tmp]\n\t"                                   /* goes on and has nothing chained: on to the stack */                         \
    "s_or_b64 %[pop], %[pop], vcc\n\t"                                                                                                                \
    "21:\n\t"                                                                                                                                          \
    "s_mov_b64 exec, %[pop]\n\t"                                        /* ---- pop; entries whose entry distance now lies beyond tHit are the ones the reference culls at :298 */ \
    "s_cbranch_execz 30f\n\t"                                                                                                                          \
    "10:\n\t"                                                                                                                                          \
    "v_cmp_ne_u32 vcc, 0, %[sp]\n\t"                                                                                                                   \
    "v_cndmask_b32 %[cur], v[GLRTX_VB+11], %[cur], vcc\n\t"                       /* empty stack: the ray is finished */                                         \
    "s_and_b64 exec, exec, vcc\n\t"                                                                                                                    \
    "s_cbranch_execz 30f\n\t"                                                                                                                          \
    "v_add_u32 %[sp], -1, %[sp]\n\t"                                                                                                                   \
    "v_lshl_add_u32 v[GLRTX_VB+15], %[sp], 11, %[stk]\n\t"                                                                                                       \
    "ds_read_b64 v[GLRTX_VB+16:GLRTX_VB+17], v[GLRTX_VB+15]\n\t"                                  /* {t0, ref} */                                                                \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                         \
    "v_mov_b32 %[cur], v[GLRTX_VB+17]\n\t"                                                                                                                       \
    "v_cmp_gt_f32 vcc, v[GLRTX_VB+16], %[th]\n\t"                                                                                                                \
    "s_and_b64 exec, exec, vcc\n\t"                                                                                                                    \
    "s_cbranch_execnz 10b\n\t"                                                                                                                         \
    "30:\n\t"                                                                                                                                          \
    "s_mov_b64 exec, %[act]\n\t"                                                                                  \
    GLRTX_TS_END                                                                                                                       \
    "v_cmpx_ne_u32_e64 %[act], %[cur], v[GLRTX_VB+11]\n\t"                        /* the lanes that go on */                                                     \
    "s_cbranch_execz 99f\n\t"

#define GLRTX_REP1(X) X
#define GLRTX_REP2(X) X X
#define GLRTX_REP3(X) X X X
#define GLRTX_REP4(X) X X X X
#define GLRTX_REP5(X) X X X X X
#define GLRTX_REP6(X) X X X X X X
#define GLRTX_REP7(X) X X X X X X X
#define GLRTX_REP8(X) X X X X X X X X
#define GLRTX_REP_(N, X) GLRTX_REP##N(X)
#define GLRTX_REP(N, X) GLRTX_REP_(N, X)
