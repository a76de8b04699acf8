// Exhaustive check: for which float32 x does a short reciprocal sequence (v_rcp_f32 + Newton steps with FMA) give exactly the
// correctly rounded 1.0f / x that the compiler's IEEE division expansion gives?  All 2^32 bit patterns.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o rcp_exact rcp_exact.hip && ./rcp_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ float rcp_hw(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float seq1(float x) { float r = rcp_hw(x); float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }
__device__ __forceinline__ float seq2(float x) { float r = seq1(x); float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }
// out[0..1]: mismatches of seq1 / seq2 inside [lo, hi]; out[2..5]: smallest / largest |x| bit pattern with a seq2 mismatch; out[6]: values in range
__global__ void k(unsigned lo_bits, unsigned hi_bits, unsigned long long *out) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long m1 = 0, m2 = 0, n = 0;
    unsigned lo_bad = 0xFFFFFFFFu, hi_bad = 0u;
    for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < (1ull << 32); u += stride) {
        const unsigned a = (unsigned)u & 0x7FFFFFFFu;
        if (a < lo_bits || a > hi_bits) continue;
        const float x = __uint_as_float((unsigned)u);
        const float ref = 1.0f / x;
        n++;
        if (__float_as_uint(seq1(x)) != __float_as_uint(ref)) m1++;
        if (__float_as_uint(seq2(x)) != __float_as_uint(ref)) { m2++; lo_bad = a < lo_bad ? a : lo_bad; hi_bad = a > hi_bad ? a : hi_bad; }
    }
    atomicAdd(&out[0], m1); atomicAdd(&out[1], m2); atomicAdd(&out[6], n);
    atomicMin((unsigned *)&out[2], lo_bad); atomicMax((unsigned *)&out[3], hi_bad);
}
int main() {
    unsigned long long *d, h[8];
    hipMalloc(&d, sizeof h);
    const float ranges[][2] = {{1e-4f, 1e30f}, {1.1754944e-38f, 1.7014118e38f}, {1e-30f, 1e30f}};
    for (auto &r : ranges) {
        unsigned lo, hi; memcpy(&lo, &r[0], 4); memcpy(&hi, &r[1], 4);
        unsigned long long init[8] = {0, 0, 0xFFFFFFFFull, 0, 0, 0, 0, 0};
        hipMemcpy(d, init, sizeof init, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, lo, hi, d);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("|x| in [%g, %g]: %llu values; rcp + 1 Newton step: %llu mismatches; rcp + 2 steps: %llu mismatches (|x| bits %08x .. %08x)\n", r[0], r[1], h[6], h[0], h[1], (unsigned)h[2], (unsigned)h[3]);
    }
    return 0;
}
