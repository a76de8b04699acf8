// Where do the workgroups of a persistent launch of the render kernel's SHAPE land, stream by stream?  (tools/gpu_ab_inproc.py: contexts of one binary run in
// two regimes 3 % apart; with every context on ONE stream they all run in the slow one -- so the regime follows the stream, i.e. the hardware queue.)
// The probe launches grid = 4 x CUs workgroups of 256 threads with 128 VGPRs and 31.5 KiB of LDS each (four per CU fit, no more) on a series of streams created
// the way glrtx_create creates them (one plain + six with priorities per "context").  Every workgroup records its XCC id, its HW_ID (SE / CU) and its start
// clock, counts itself in, and waits until every workgroup of the grid has arrived or 4 ms have passed.
// Output per stream: workgroups resident together, workgroups per XCC (min .. max), per CU (histogram of 0..5+), spread of the start clocks.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>
struct Rec { unsigned xcc, hwid; unsigned long long t0; unsigned seen, pad; };
__global__ __launch_bounds__(256, 4) void k(Rec *rec, unsigned *arrived, unsigned grid) {
    extern __shared__ unsigned char lds[];
    asm volatile("v_mov_b32 v127, 0" ::: "v127");  // 128 VGPRs per wave, like the render kernel
    if (threadIdx.x == 0) {
        lds[0] = 1;
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(arrived, 1u);
        unsigned seen = 0;
        while (__builtin_amdgcn_s_memrealtime() - t0 < 400000ull) {  // 100 MHz: 4 ms
            seen = __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen >= grid) break;
            __builtin_amdgcn_s_sleep(32);
        }
        rec[blockIdx.x] = {xcc & 0xFu, hw, t0, seen, 0u};
    }
    __syncthreads();
}
int main(int argc, char **argv) {
    const int n_ctx = argc > 1 ? atoi(argv[1]) : 6;
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const unsigned grid = 4u * (unsigned)pr.multiProcessorCount;
    const int lds = 32256;
    Rec *d; unsigned *arr;
    hipMalloc(&d, grid * sizeof(Rec)); hipMalloc(&arr, 4);
    int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("%d CUs, grid %u, stream priorities %d (least) .. %d (greatest)\n", pr.multiProcessorCount, grid, lo, hi);
    std::vector<hipStream_t> keep;
    for (int c = 0; c < n_ctx; c++) {
        hipStream_t own; hipStreamCreateWithFlags(&own, hipStreamNonBlocking); keep.push_back(own);
        for (int s = 0; s < 6; s++) { hipStream_t t; hipStreamCreateWithPriority(&t, hipStreamNonBlocking, hi); keep.push_back(t); }  // the pipe slots' streams
        for (int rep = 0; rep < 2; rep++) {
            hipMemsetAsync(arr, 0, 4, own);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, own, d, arr, grid);
            if (hipStreamSynchronize(own) != hipSuccess) { printf("launch failed\n"); return 2; }
            std::vector<Rec> h(grid); hipMemcpy(h.data(), d, grid * sizeof(Rec), hipMemcpyDeviceToHost);
            unsigned per_xcc[16] = {0}, seen_max = 0; std::map<unsigned long long, int> per_cu; unsigned long long tmin = ~0ull, tmax = 0;
            for (auto &r : h) {
                per_xcc[r.xcc]++; seen_max = std::max(seen_max, r.seen);
                const unsigned cu = r.hwid >> 8 & 0xFu, sh = r.hwid >> 12 & 1u, se = r.hwid >> 13 & 7u;  // HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
                per_cu[(unsigned long long)r.xcc << 16 | se << 8 | sh << 4 | cu]++;
                tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t0);
            }
            int hist[8] = {0}; for (auto &kv : per_cu) hist[std::min(kv.second, 7)]++;
            unsigned xmin = ~0u, xmax = 0; for (int x = 0; x < 8; x++) { xmin = std::min(xmin, per_xcc[x]); xmax = std::max(xmax, per_xcc[x]); }
            printf("context %d launch %d: resident together %u of %u; per XCC %u .. %u; CUs seen %zu, workgroups per CU: 1:%d 2:%d 3:%d 4:%d 5+:%d; starts spread over %.2f us\n", c, rep, seen_max, grid,
                   xmin, xmax, per_cu.size(), hist[1], hist[2], hist[3], hist[4], hist[5] + hist[6] + hist[7], (tmax - tmin) / 100.0);
        }
    }
    return 0;
}
