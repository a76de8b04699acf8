// Exhaustive check of the short two-operand quotients of tools/ubench/quotient.hip.h (an experiment of round 4: exact, not faster, not in the kernels --
// profiles/r04_ab_short_fdiv.txt), on the device, in the
// kernels' float mode (fp32 denormals flushed on input and output), against the compiler's correctly rounded IEEE division.
//   part 1  every pair of significands: a = 1.ma, b = 1.mb for all 2^23 x 2^23 (ma, mb): quot_step(a, b, quot_rcp(b)) == a / b bit for bit.  With the three
//           range tests of quotient.hip.h passed nothing on the way leaves the normal range, so other exponents only shift every intermediate value exactly.
//   part 2  the range tests themselves: for every pair of exponent fields (256 x 256), both signs and 32 x 32 significands (the edges 0, 1, 2^22, 2^23 - 1, ...
//           and random ones), and for 2^36 random pairs of bit patterns: a lane that the tests let through holds a / b bit for bit.  (A lane they do not
//           let through takes a / b itself; here such lanes are only counted.)
//   part 3  fdiv / fdiv3 as the kernels call them (wave-level decision) on the random pairs: == a / b, NaNs as NaNs.
// Usage: div_exact [--quick]   (--quick: part 1 on every 64th divisor significand, 2^40 pairs, and 2^32 random pairs)
// Built like the kernels:  cd tools/ubench && hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include "quotient.hip.h"

DEV bool same_bits(float x, float y) { return __float_as_uint(x) == __float_as_uint(y); }
DEV bool same_or_nan(float x, float y) { return same_bits(x, y) || (x != x && y != y); }

// part 1: thread = one divisor significand (mb0 + global id) * mb_stride, loop over numerator significands [ma0, ma0 + ma_n)
__global__ void k_significands(unsigned long long *out, unsigned mb0, unsigned mb_stride, unsigned ma0, unsigned ma_n) {
    const unsigned mb = (mb0 + blockIdx.x * blockDim.x + threadIdx.x) * mb_stride;
    if (mb >= (1u << 23)) return;
    const float b = __uint_as_float(0x3F800000u | mb);
    const float r = quot_rcp(b);
    unsigned long long bad = 0, n = 0;
    for (unsigned ma = ma0; ma < ma0 + ma_n; ma++) {
        const float a = __uint_as_float(0x3F800000u | ma);
        bool outside = quot_divisor_outside(b);
        const float s = quot_step(a, b, r, outside);
        if (outside || !same_bits(s, a / b)) bad++;
        n++;
    }
    if (bad) atomicAdd(&out[0], bad);
    atomicAdd(&out[1], n);
}

DEV unsigned edge_significand(unsigned i, unsigned salt) {
    switch (i) {
        case 0: return 0u;
        case 1: return 1u;
        case 2: return 0x7FFFFFu;
        case 3: return 0x7FFFFEu;
        case 4: return 0x400000u;
        case 5: return 0x3FFFFFu;
        case 6: return 0x400001u;
        case 7: return 0x000002u;
        default: {
            unsigned x = (i + 1u) * 0x9E3779B9u ^ salt * 0x85EBCA6Bu;
            x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
            return x & 0x7FFFFFu;
        }
    }
}
DEV void judge(float a, float b, unsigned long long *m) {
    bool outside = quot_divisor_outside(b);
    const float s = quot_step(a, b, quot_rcp(b), outside);
    if (outside) m[1]++;
    else if (!same_bits(s, a / b)) m[0]++;
    m[2]++;
}
// part 2a: block = (ea, eb), thread = (significand choice of a, of b) for the four sign pairs
__global__ void k_exponents(unsigned long long *out) {
    const unsigned ea = blockIdx.x & 255u, eb = blockIdx.x >> 8;
    unsigned long long m[3] = {0, 0, 0};
    for (unsigned t = threadIdx.x; t < 1024u; t += blockDim.x) {
        const unsigned ma = edge_significand(t & 31u, ea * 256u + eb), mb = edge_significand(t >> 5, eb * 256u + ea + 77u);
        for (unsigned sg = 0; sg < 4u; sg++)
            judge(__uint_as_float((sg & 1u) << 31 | ea << 23 | ma), __uint_as_float((sg >> 1) << 31 | eb << 23 | mb), m);
    }
    for (int i = 0; i < 3; i++) atomicAdd(&out[2 + i], m[i]);
}
DEV unsigned long long splitmix(unsigned long long &x) {
    unsigned long long z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// part 2b + 3: random pairs of bit patterns; a quarter of them with the exponent fields drawn near the range tests' thresholds
__global__ void k_random(unsigned long long *out, unsigned long long seed, unsigned per_thread) {
    unsigned long long st = seed + 0x632BE59BD9B4E019ull * ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x + 1ull);
    unsigned long long m[3] = {0, 0, 0}, w = 0;
    for (unsigned i = 0; i < per_thread; i++) {
        const unsigned long long z = splitmix(st);
        unsigned ua = (unsigned)z, ub = (unsigned)(z >> 32);
        if ((i & 3u) == 3u) {  // numerators around 2^-79 and 0, divisors around 2^40, quotients around the ends of the range
            const unsigned long long y = splitmix(st);
            const unsigned ea = (y & 1u) ? 127u - 79u + (unsigned)((y >> 1) % 5u) - 2u : (unsigned)((y >> 1) & 255u);
            const unsigned eb = (y & 2u) ? 127u + 40u + (unsigned)((y >> 9) % 5u) - 2u : (unsigned)((y >> 9) & 255u);
            ua = (ua & 0x807FFFFFu) | ea << 23; ub = (ub & 0x807FFFFFu) | eb << 23;
        }
        const float a = __uint_as_float(ua), b = __uint_as_float(ub);
        judge(a, b, m);
        // the kernels' own entry points (wave-level decision)
        if (!same_or_nan(fdiv(a, b), a / b)) w++;
        float x0 = a, x1 = __uint_as_float(ua ^ 0x00012345u), x2 = __uint_as_float(ub ^ 0x40000000u);
        const float y0 = x0 / b, y1 = x1 / b, y2 = x2 / b;
        fdiv3(x0, x1, x2, b);
        if (!same_or_nan(x0, y0) || !same_or_nan(x1, y1) || !same_or_nan(x2, y2)) w++;
    }
    for (int i = 0; i < 3; i++) atomicAdd(&out[5 + i], m[i]);
    atomicAdd(&out[8], w);
}

static bool sync_ok(const char *what) {
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); return false; }
    return true;
}
int main(int argc, char **argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "--quick");
    unsigned long long *d, h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // [0] significand mismatches [1] pairs | [2] exponent-sweep mismatches [3] outside [4] pairs | [5..7] the same for random pairs | [8] entry-point mismatches
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 2;
    if (hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice) != hipSuccess) return 2;
    // part 1: 2^23 divisor significands (every 64th with --quick) x 2^23 numerator significands, in slices of 2^18 numerators per launch
    const unsigned stride = quick ? 64u : 1u, n_mb = (1u << 23) / stride, slice = 1u << 18;
    for (unsigned ma0 = 0; ma0 < (1u << 23); ma0 += slice) {
        hipLaunchKernelGGL(k_significands, dim3((n_mb + 255u) / 256u), dim3(256), 0, 0, d, 0u, stride, ma0, slice);
        if (!sync_ok("k_significands")) return 2;
        if ((ma0 / slice) % 8u == 7u) { printf("significands: %u of 32 slices done\n", ma0 / slice + 1u); fflush(stdout); }
    }
    hipLaunchKernelGGL(k_exponents, dim3(65536), dim3(256), 0, 0, d);
    if (!sync_ok("k_exponents")) return 2;
    const unsigned per_thread = quick ? 1024u : 16384u;  // 4096 x 1024 threads x per_thread = 2^32 / 2^36 pairs
    for (unsigned part = 0; part < 4u; part++) {
        hipLaunchKernelGGL(k_random, dim3(4096), dim3(256), 0, 0, d + 0, 0x1234ull + part, per_thread);
        if (!sync_ok("k_random")) return 2;
    }
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    printf("significand pairs %llu: %llu mismatches\n", h[1], h[0]);
    printf("exponent sweep %llu pairs: %llu outside the range tests, %llu mismatches among the others\n", h[4], h[3], h[2]);
    printf("random pairs %llu: %llu outside the range tests, %llu mismatches among the others\n", h[7], h[6], h[5]);
    printf("fdiv / fdiv3 on the random pairs: %llu mismatches\n", h[8]);
    return (h[0] | h[2] | h[5] | h[8]) ? 1 : 0;
}
