// Micro-benchmark: dependent 64-byte record gathers, the memory pattern of the BVH traversal loop.
//   mode 0: each lane reads its own record with 4 x global_load_dwordx4 (what trav_step does)
//   mode 1: each lane reads only 16 B of its record (lower bound: one line lookup per lane-step)
//   mode 2: quad-cooperative: in instruction k the 4 lanes of a quad read the 4 chunks of lane k's record
//           (one 64-B coalesced request per quad), then a DPP transpose hands every lane its own record
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/gather.hip -o tools/ubench/gather ; run: tools/ubench/gather [log2 records] [json-lines file]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned a, unsigned b) { return (a * 0x9E3779B1u) ^ (b + 0x7F4A7C15u) ^ (a >> 15); }

template <int C>
__device__ __forceinline__ unsigned quad_bcast(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, C | (C << 2) | (C << 4) | (C << 6), 0xF, 0xF, true);
}
template <int C, int J>
__device__ __forceinline__ void quad_put(unsigned &dst, unsigned src) {
    // lanes with (lane & 3) == J receive src of quad lane C; the others keep dst
    dst = (unsigned)__builtin_amdgcn_update_dpp((int)dst, (int)src, C | (C << 2) | (C << 4) | (C << 6), 0xF, 1 << J, false);
}

template <int MODE>
__global__ __launch_bounds__(256, 4) void chase(const uint4 *__restrict__ rec, unsigned n_mask, int steps, unsigned *out) {
    unsigned cur = mix(blockIdx.x * 256 + threadIdx.x, 12345u) & n_mask;
    unsigned acc = 0;
    const unsigned lane4 = threadIdx.x & 3;
    for (int s = 0; s < steps; s++) {
        uint4 A, B, Cc, D;
        if (MODE == 0) {
            const uint4 *p = rec + 4 * (size_t)cur;
            A = p[0]; B = p[1]; Cc = p[2]; D = p[3];
        } else if (MODE == 1) {
            A = rec[4 * (size_t)cur];
            B = A; Cc = A; D = A;
        } else if (MODE == 3) {  // one 16-B load, non-temporal hint
            const uint4 *p = rec + 4 * (size_t)cur;
            asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(A) : "v"(p) : "memory");
            B = A; Cc = A; D = A;
        } else if (MODE == 4) {  // one 16-B load, sc0 sc1 (bypass / coherent at system scope)
            const uint4 *p = rec + 4 * (size_t)cur;
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(A) : "v"(p) : "memory");
            B = A; Cc = A; D = A;
        } else if (MODE == 5) {  // one 16-B load; lane pairs share a 128-B line (different 64-B halves)
            const unsigned c2 = ((unsigned)__shfl((int)cur, (threadIdx.x & 63) & ~1) & ~1u) | (threadIdx.x & 1u);
            A = rec[4 * (size_t)c2];
            B = A; Cc = A; D = A;
        } else if (MODE == 6) {  // one 4-B load
            const unsigned v = reinterpret_cast<const unsigned *>(rec)[16 * (size_t)cur];
            A = make_uint4(v, v * 3u, v * 5u, v * 7u);
            B = A; Cc = A; D = A;
        } else if (MODE == 7) {  // four 16-B loads, sc1 (L1 policy hint) -- same bytes as mode 0
            const uint4 *p = rec + 4 * (size_t)cur;
            asm volatile("global_load_dwordx4 %0, %4, off nt\n\tglobal_load_dwordx4 %1, %4, off offset:16 nt\n\t"
                         "global_load_dwordx4 %2, %4, off offset:32 nt\n\tglobal_load_dwordx4 %3, %4, off offset:48 nt\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(A), "=&v"(B), "=&v"(Cc), "=&v"(D) : "v"(p) : "memory");
        } else if (MODE == 8) {  // lanes of a quad share one record (4x fewer distinct lines): coherent-ray limit
            const unsigned c4 = (unsigned)__shfl((int)cur, (threadIdx.x & 63) & ~3);
            const uint4 *p = rec + 4 * (size_t)c4;
            A = p[0]; B = p[1]; Cc = p[2]; D = p[3];
            A.x += threadIdx.x & 3;
        } else {
            const unsigned c0 = quad_bcast<0>(cur), c1 = quad_bcast<1>(cur), c2 = quad_bcast<2>(cur), c3 = quad_bcast<3>(cur);
            const uint4 R0 = rec[4 * (size_t)c0 + lane4];
            const uint4 R1 = rec[4 * (size_t)c1 + lane4];
            const uint4 R2 = rec[4 * (size_t)c2 + lane4];
            const uint4 R3 = rec[4 * (size_t)c3 + lane4];
            // lane j needs chunk c of its record = R_j held by quad lane c
#define TR(OUT, CH)                                                                         \
            quad_put<CH, 0>(OUT.x, R0.x); quad_put<CH, 1>(OUT.x, R1.x); quad_put<CH, 2>(OUT.x, R2.x); quad_put<CH, 3>(OUT.x, R3.x); \
            quad_put<CH, 0>(OUT.y, R0.y); quad_put<CH, 1>(OUT.y, R1.y); quad_put<CH, 2>(OUT.y, R2.y); quad_put<CH, 3>(OUT.y, R3.y); \
            quad_put<CH, 0>(OUT.z, R0.z); quad_put<CH, 1>(OUT.z, R1.z); quad_put<CH, 2>(OUT.z, R2.z); quad_put<CH, 3>(OUT.z, R3.z); \
            quad_put<CH, 0>(OUT.w, R0.w); quad_put<CH, 1>(OUT.w, R1.w); quad_put<CH, 2>(OUT.w, R2.w); quad_put<CH, 3>(OUT.w, R3.w);
            A = B = Cc = D = make_uint4(0, 0, 0, 0);
            TR(A, 0) TR(B, 1) TR(Cc, 2) TR(D, 3)
#undef TR
        }
        const unsigned h = (A.x ^ B.y ^ Cc.z ^ D.w) + (A.y ^ B.z ^ Cc.w ^ D.x) * 3u + (A.z ^ B.w ^ Cc.x ^ D.y) * 5u + (A.w ^ B.x ^ Cc.y ^ D.z) * 7u;
        acc += h;
        cur = mix(h, cur + s) & n_mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const int log_n = argc > 1 ? atoi(argv[1]) : 15;  // 2^15 records * 64 B = 2 MB
    const unsigned n = 1u << log_n;
    const int steps = 256, blocks = 4096;
    std::vector<unsigned> h((size_t)n * 16);
    unsigned x = 1;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = x; }
    uint4 *rec; unsigned *out;
    CK(hipMalloc(&rec, h.size() * 4)); CK(hipMalloc(&out, blocks * 256 * 4));
    CK(hipMemcpy(rec, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned> r0(blocks * 256), r2(blocks * 256);
    static const char *what[9] = {"4 x dwordx4 of the lane's own 64-B record (one 128-B line per lane)", "one dwordx4 of the record", "quad-cooperative 64-B loads + DPP transpose",
                                  "one dwordx4, nt", "one dwordx4, sc0 sc1", "one dwordx4, lane pairs share a 128-B line", "one dword", "4 x dwordx4, nt",
                                  "4 x dwordx4, the 4 lanes of a quad share one record (16 distinct lines per wave)"};
    FILE *js = argc > 2 ? fopen(argv[2], "a") : nullptr;
    for (int mode = 0; mode < 9; mode++) {
        float best = 1e9f;
        for (int it = 0; it < 4; it++) {
            CK(hipEventRecord(e0));
            if (mode == 0) chase<0><<<blocks, 256>>>(rec, n - 1, steps, out);
            if (mode == 1) chase<1><<<blocks, 256>>>(rec, n - 1, steps, out);
            if (mode == 2) chase<2><<<blocks, 256>>>(rec, n - 1, steps, out);
            if (mode == 3) chase<3><<<blocks, 256>>>(rec, n - 1, steps, out);
            if (mode == 4) chase<4><<<blocks, 256>>>(rec, n - 1, steps, out);
            if (mode == 5) chase<5><<<blocks, 256>>>(rec, n - 1, steps, out);
            if (mode == 6) chase<6><<<blocks, 256>>>(rec, n - 1, steps, out);
            if (mode == 7) chase<7><<<blocks, 256>>>(rec, n - 1, steps, out);
            if (mode == 8) chase<8><<<blocks, 256>>>(rec, n - 1, steps, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        if (mode == 0) CK(hipMemcpy(r0.data(), out, r0.size() * 4, hipMemcpyDeviceToHost));
        if (mode == 2) CK(hipMemcpy(r2.data(), out, r2.size() * 4, hipMemcpyDeviceToHost));
        const double lane_steps = (double)blocks * 256 * steps;
        printf("mode %d records 2^%d: %.3f ms, %.2f G lane-steps/s, %.3f lane-steps/clk/CU @2.4GHz\n", mode, log_n, best,
               lane_steps / best * 1e-6, lane_steps / (best * 1e-3) / 256 / 2.4e9);
        if (js) fprintf(js, "{\"mode\": %d, \"what\": \"%s\", \"table_bytes\": %zu, \"ms\": %.4f, \"g_lane_steps_per_s\": %.3f, \"clk_per_wave_step_per_cu\": %.2f}\n", mode, what[mode],
                        (size_t)n * 64, best, lane_steps / best * 1e-6, 64.0 / (lane_steps / (best * 1e-3) / 256 / 2.4e9));
    }
    printf("mode2 == mode0: %s\n", r0 == r2 ? "yes" : "NO");
    if (js) fclose(js);
    return 0;
}
