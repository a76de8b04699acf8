// Two-operand quotients a / b of the shade phase (raytrace.frag:158-219, :337-403, :546, :553) without the compiler's IEEE expansion.
// EXPERIMENT RECORD (round 4): exact on the device, not faster in the render kernel, so pt_kernel.hip.h keeps the compiler's quotient (profiles/r04_ab_short_fdiv.txt).
//
// In the kernels' float mode (fp32 denormals flushed) the compiler's correctly rounded a / b is v_div_scale x 2, v_rcp, five fused
// multiply-adds, v_div_fmas, v_div_fixup and two s_setreg that switch denormals on and off around the core: 14 vector + 2 scalar instructions per
// quotient, ten quotients per diffuse bounce, twenty-two more in the conductor branch.  The short form is
//     r = RN(1 / b)            v_rcp_f32 + one Newton step (rcp_newton: the IEEE reciprocal of every normal float, tools/ubench/rcp_exact.hip)
//     q = a * r,  e = fma(-b, q, a),  s = fma(e, r, q)
// which is the correctly rounded quotient whenever nothing on the way leaves the normal range (Markstein's correction step with a correctly
// rounded reciprocal; tools/ubench/div_exact.hip runs ALL 2^46 pairs of significands through it on the device against the compiler's quotient:
// the operands' exponents only shift every intermediate value exactly).  Three tests tell whether anything can leave the range:
//     s is a normal number (v_cmp_class: not zero, denormal, infinite or NaN),   |a| >= 2^-79,   |b| <= 2^40.
//   * b zero, denormal, infinite or NaN: r is inf, 0 or NaN and s comes out as NaN or 0 -- not normal;
//   * a infinite or NaN: e is NaN, so is s;  a quotient that overflows: s is inf or NaN;  a zero numerator: s = 0 (with the wrong sign for a = -0, b > 0);
//   * the residual e is a multiple of 2^(exponent(a) - 47): with |a| >= 2^-79 it is zero or normal, never flushed;
//   * |a / b| >= 2^-120 with these bounds: q and s stay clear of the lowest binade, where flushing decides.
// A wave with any lane outside takes the compiler's quotient (a branch, not a select between both); the same tool checks that decision on every pair of
// exponents and on 2^36 random pairs of bit patterns.  Quotients that share a divisor share r (fdiv3: beta *= f cos / pdf, the light sample, the roulette).
#pragma once
#ifndef DEV
#define DEV __device__ __forceinline__
#endif

DEV float quot_rcp(float b) {  // == rcp_newton of pt_kernel.hip.h
    const float r = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
}
DEV float quot_step(float a, float b, float r, bool &outside) {
    const float q = a * r;
    const float s = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
    outside = outside || !__builtin_amdgcn_class(s, 0x108) || !(__builtin_fabsf(a) >= 0x1p-79f);  // 0x108: -normal | +normal
    return s;
}
DEV bool quot_divisor_outside(float b) { return !(__builtin_fabsf(b) <= 0x1p40f); }
#ifdef GLRTX_IEEE_FDIV  // A/B switch: the compiler's quotient everywhere
DEV float fdiv(float a, float b) { return a / b; }
DEV void fdiv3(float &a0, float &a1, float &a2, float b) { a0 = a0 / b; a1 = a1 / b; a2 = a2 / b; }
#else
DEV float fdiv(float a, float b) {
#ifdef GLRTX_IEEE_FDIV1
    return a / b;
#endif
    bool outside = quot_divisor_outside(b);
    const float s = quot_step(a, b, quot_rcp(b), outside);
    if (__any(outside)) return a / b;
    return s;
}
// a0 / b, a1 / b, a2 / b in place
DEV void fdiv3(float &a0, float &a1, float &a2, float b) {
    bool outside = quot_divisor_outside(b);
    const float r = quot_rcp(b);
    const float s0 = quot_step(a0, b, r, outside), s1 = quot_step(a1, b, r, outside), s2 = quot_step(a2, b, r, outside);
    if (__any(outside)) { a0 = a0 / b; a1 = a1 / b; a2 = a2 / b; return; }
    a0 = s0; a1 = s1; a2 = s2;
}
#endif
