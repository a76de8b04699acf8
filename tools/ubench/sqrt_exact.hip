// Exhaustive check (all 2^32 bit patterns): which short sequences equal the correctly rounded sqrtf(x) of the compiler's expansion
// (-fhip-fp32-correctly-rounded-divide-sqrt), and where?
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o sqrt_exact sqrt_exact.hip && ./sqrt_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ float seqA(float x) {  // v_sqrt, v_rcp, one Heron correction
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rcpf(s);
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
__device__ __forceinline__ float seqB(float x) {  // v_rsq, Goldschmidt-style correction
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y, h = 0.5f * y;
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
__device__ __forceinline__ float seqC(float x) { return __builtin_amdgcn_sqrtf(x); }  // the instruction alone
__global__ void k(unsigned lo_bits, unsigned hi_bits, unsigned long long *out) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long m[3] = {0, 0, 0}, n = 0;
    for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < (1ull << 31); u += stride) {  // x >= 0
        if ((unsigned)u < lo_bits || (unsigned)u > hi_bits) continue;
        const float x = __uint_as_float((unsigned)u);
        const unsigned ref = __float_as_uint(__builtin_sqrtf(x));
        n++;
        m[0] += __float_as_uint(seqA(x)) != ref;
        m[1] += __float_as_uint(seqB(x)) != ref;
        m[2] += __float_as_uint(seqC(x)) != ref;
    }
    atomicAdd(&out[0], m[0]); atomicAdd(&out[1], m[1]); atomicAdd(&out[2], m[2]); atomicAdd(&out[3], n);
}
int main() {
    unsigned long long *d, h[4];
    (void)hipMalloc(&d, sizeof h);
    const float ranges[][2] = {{1e-30f, 1e30f}, {1.1754944e-38f, 3.4028235e38f}, {1e-36f, 1e36f}};
    for (auto &r : ranges) {
        unsigned lo, hi; memcpy(&lo, &r[0], 4); memcpy(&hi, &r[1], 4);
        (void)hipMemset(d, 0, sizeof h);
        hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, lo, hi, d);
        (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("x in [%g, %g]: %llu values; mismatches: sqrt+rcp+Heron %llu, rsq form %llu, v_sqrt_f32 alone %llu\n", r[0], r[1], h[3], h[0], h[1], h[2]);
    }
    return 0;
}
