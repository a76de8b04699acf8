// Exhaustive checks behind the short quotients of pt_kernel.hip.h, trav_asm.hip.h and scan_asm.hip.h: every float32 bit pattern, on the device, against the
// compiler's correctly rounded IEEE division -- in the float mode the kernels run in: fp32 denormals flushed on input and output (-fgpu-flush-denormals-to-zero,
// what the reference's GL implementation does: llvmpipe's rasteriser threads set MXCSR FTZ | DAZ).
//   newton(x)  = v_rcp_f32 + one Newton step with fused multiply-adds                     -- 1 / det in the triangle tests (no guard around it)
//   frcp(x)    = newton(x), but the raw v_rcp_f32 result where x is a zero, a denormal or an infinity   -- 1 / x everywhere else (rsq, 1 / d, ...)
//   div_pi(x)  = x * RN(1 / PI) corrected by one residual step; full division for the waves that hold a tiny or huge x       -- x / PI
// Built like the kernels:  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define DEV __device__ __forceinline__
constexpr float PT_PI = 3.14159274101257324f;
constexpr float PT_INV_PI = 0.318309873342514038f;  // RN(1 / PT_PI)
DEV float newton(float x) { const float r = __builtin_amdgcn_rcpf(x); return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r); }
DEV float frcp(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    const float n = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
    return __builtin_amdgcn_class(x, 0x2F4) ? r : n;  // -inf, -denormal, -0, +0, +denormal, +inf
}
DEV float div_pi(float x) {
    const unsigned a = __float_as_uint(x) & 0x7FFFFFFFu;
    const bool plain = __float_as_uint(x) != 0u && (a - 0x0D800000u) > (0x7B800000u - 0x0D800000u);  // -0, 0 < |x| < 2^-100, |x| > 2^120, inf, NaN
    if (__any(plain)) return x / PT_PI;
    const float q = x * PT_INV_PI;
    return __builtin_fmaf(__builtin_fmaf(-PT_PI, q, x), PT_INV_PI, q);
}
// the short form by itself, and its per-lane range predicate: checked on EVERY in-range pattern -- div_pi() above takes the full division for the whole wave as soon as
// one of its 64 consecutive patterns is out of range, so the wave that holds the range's inclusive ends (|x| = 2^-100, 2^120) never runs the short form there
DEV float div_pi_short(float x) { const float q = x * PT_INV_PI; return __builtin_fmaf(__builtin_fmaf(-PT_PI, q, x), PT_INV_PI, q); }
DEV bool div_pi_in_range(float x) { const unsigned a = __float_as_uint(x) & 0x7FFFFFFFu; return !(__float_as_uint(x) != 0u && (a - 0x0D800000u) > (0x7B800000u - 0x0D800000u)); }
DEV bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
// out[6], [7]: mismatches of div_pi's short form evaluated unconditionally / patterns inside its range.  out[0]: newton mismatches with |x| >= FLT_MIN and finite; [1]: newton mismatches among zeros / denormals / infinities / NaNs; [2]: frcp mismatches (all);
// [3]: div_pi mismatches (all); [4]: patterns visited; [5]: newton mismatches with |x| > 2^126 (the quotient is flushed: both must give a zero of det's sign)
__global__ void k(unsigned long long *out) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < (1ull << 32); u += stride) {
        const float x = __uint_as_float((unsigned)u);
        const unsigned a = (unsigned)u & 0x7FFFFFFFu;
        const float ref = 1.0f / x;
        const bool ordinary = a >= 0x00800000u && a < 0x7F800000u;
        if (!same(newton(x), ref)) { m[ordinary ? 0 : 1]++; if (ordinary && a > 0x7E800000u) m[5]++; }
        if (!same(frcp(x), ref)) m[2]++;
        if (!same(div_pi(x), x / PT_PI)) m[3]++;
        if (div_pi_in_range(x)) { m[7]++; if (!same(div_pi_short(x), x / PT_PI)) m[6]++; }
        m[4]++;
    }
    for (int i = 0; i < 8; i++) atomicAdd(&out[i], m[i]);
}
int main() {
    unsigned long long *d, h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 2;
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    printf("patterns %llu\n", h[4]);
    printf("newton: %llu mismatches among normal finite x (%llu of them above 2^126), %llu among zeros, denormals, infinities and NaNs\n", h[0], h[5], h[1]);
    printf("frcp: %llu mismatches\n", h[2]);
    printf("div_pi: %llu mismatches\n", h[3]);
    printf("div_pi short form alone: %llu mismatches among the %llu patterns of its range (ends included)\n", h[6], h[7]);
    return 0;
}
