// Micro-benchmark: what a CU's vector-memory pipe (address unit + L1) charges for a wave's global_load instruction, by width, by
// the number of lanes enabled, and by the number of distinct 128-byte lines the lanes touch -- on L1-resident data, so that nothing
// behind the L1 is measured.  The traversal step of the render kernel issues four such loads (a 56-byte record per lane) and the
// frame time moves by ~10 % per load instruction added to the step (profiles/r03_traverse_bound.txt): this prices the instruction.
//   4 workgroups x 256 threads per CU (4 waves per SIMD, the render kernel's occupancy); every wave issues ROUNDS x 16 loads of one kind
//   with s_waitcnt vmcnt(0) after each group of 16; reported: clocks per wave-instruction per CU = elapsed / (wave-instructions per CU)
//   at the clock s_memtime counts (shader clock), and chip-wide G wave-instructions/s.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/ta.hip -o tools/ubench/ta ; run: tools/ubench/ta [json-lines file]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f3 __attribute__((ext_vector_type(3)));
template <int W> struct Reg { typedef f4 T; };
template <> struct Reg<3> { typedef f3 T; };
template <> struct Reg<2> { typedef f2 T; };
template <> struct Reg<1> { typedef float T; };
static __device__ __forceinline__ float first(float v) { return v; }
template <class V> static __device__ __forceinline__ float first(V v) { return v.x; }

// WIDTH: dwords per lane (1, 2, 3 = dwordx3, 4).  The address of a lane is base + line_of_lane * 128 + (lane & 7) * 16: `lines` distinct
// lines per wave instruction (1, 8, 16, 64); all addresses stay inside a 64 x 128 B = 8 KiB window per wave -> L1-resident after the first touch.
template <int WIDTH>
__global__ __launch_bounds__(256, 4) void ta_kernel(const char *table, int rounds, int lines, unsigned long long exec_mask, float *out, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6)) & 63;
    const int line = lines >= 64 ? lane : (lines <= 1 ? 0 : lane % lines);
    const char *p = table + (size_t)wave * 8192 + (size_t)line * 128 + (lane & 7) * 16;
    float acc = 0.f;
    const bool on = (exec_mask >> lane) & 1ull;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (on) {
        for (int r = 0; r < rounds; r++) {
            typename Reg<WIDTH>::T v0, v1, v2, v3;
#define L4(OP, V, OFF) asm volatile(OP " %0, %1, off offset:" #OFF : "=v"(V) : "v"(p) : "memory")
#define GROUP(OP)                                                                                                     \
            L4(OP, v0, 0); L4(OP, v1, 0); L4(OP, v2, 0); L4(OP, v3, 0); L4(OP, v0, 0); L4(OP, v1, 0); L4(OP, v2, 0); L4(OP, v3, 0); \
            L4(OP, v0, 0); L4(OP, v1, 0); L4(OP, v2, 0); L4(OP, v3, 0); L4(OP, v0, 0); L4(OP, v1, 0); L4(OP, v2, 0); L4(OP, v3, 0);
            if constexpr (WIDTH == 4) { GROUP("global_load_dwordx4") }
            else if constexpr (WIDTH == 3) { GROUP("global_load_dwordx3") }
            else if constexpr (WIDTH == 2) { GROUP("global_load_dwordx2") }
            else { GROUP("global_load_dword") }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "memory");
            acc += first(v0) + first(v1) + first(v2) + first(v3);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main(int argc, char **argv) {
    FILE *js = argc > 1 ? fopen(argv[1], "a") : nullptr;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount, blocks = n_cu * 4, rounds = 512;
    char *table; float *out; unsigned long long *cyc;
    CK(hipMalloc(&table, 64 * 8192 + 4096)); CK(hipMemset(table, 0, 64 * 8192 + 4096));
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&cyc, (size_t)blocks * 4 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Mask { const char *name; unsigned long long m; int active; };
    const Mask masks[] = {{"all 64 lanes", ~0ull, 64}, {"lanes 0-47", 0x0000FFFFFFFFFFFFull, 48}, {"lanes 0-31", 0x00000000FFFFFFFFull, 32}, {"lanes 0-15", 0xFFFFull, 16},
                          {"every other lane (32)", 0x5555555555555555ull, 32}, {"every other quad (32)", 0x0F0F0F0F0F0F0F0Full, 32}, {"one lane per quad (16)", 0x1111111111111111ull, 16}};
    const int line_counts[] = {1, 16, 64};
    printf("%d CUs; 4 workgroups x 4 waves per CU\n", n_cu);
    for (int width = 1; width <= 4; width++)
        for (const Mask &mk : masks)
            for (int lines : line_counts) {
                if (mk.active != 64 && lines == 16) continue;
                float best = 1e9f; double clk = 0;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipEventRecord(e0));
                    if (width == 4) ta_kernel<4><<<blocks, 256>>>(table, rounds, lines, mk.m, out, cyc);
                    if (width == 3) ta_kernel<3><<<blocks, 256>>>(table, rounds, lines, mk.m, out, cyc);
                    if (width == 2) ta_kernel<2><<<blocks, 256>>>(table, rounds, lines, mk.m, out, cyc);
                    if (width == 1) ta_kernel<1><<<blocks, 256>>>(table, rounds, lines, mk.m, out, cyc);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) {
                        best = ms;
                        std::vector<unsigned long long> h((size_t)blocks * 4);
                        CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
                        double s = 0; for (auto v : h) s += (double)v;
                        clk = s / h.size();  // mean clocks a wave spent in the loop; 16 waves per CU run side by side
                    }
                }
                const double inst_per_cu = 16.0 * rounds * 16;  // wave-instructions issued by one CU's 16 waves
                printf("dwordx%d  %-24s %2d line(s): %6.2f clk per wave-instruction per CU, %7.1f G wave-inst/s chip-wide (%.3f ms)\n", width, mk.name, lines,
                       clk / (rounds * 16.0) / 16.0, inst_per_cu * n_cu / (best * 1e-3) * 1e-9, best);
                if (js) fprintf(js, "{\"width_dwords\": %d, \"lanes\": \"%s\", \"active_lanes\": %d, \"distinct_lines\": %d, \"clk_per_wave_inst_per_cu\": %.3f, \"g_wave_inst_per_s\": %.2f, \"ms\": %.4f}\n",
                                width, mk.name, mk.active, lines, clk / (rounds * 16.0) / 16.0, inst_per_cu * n_cu / (best * 1e-3) * 1e-9, best);
            }
    if (js) fclose(js);
    return 0;
}
