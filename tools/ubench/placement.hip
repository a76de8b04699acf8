// Is the render kernel's "fast / slow context" lottery (profiles/r04_context_regimes.txt: it follows the placement of the 4 GB path-state buffer alone) visible to a simple
// kernel?  N buffers of the path-state size from hipMalloc, all alive together; on each: a streaming read, a streaming write, and a read-modify-write of four of six planes in
// the renderer's pattern (1024 workgroups, each walking its own windows of 4096 consecutive ids through planes that lie `ids` entries apart).  A metric that splits the buffers
// into two groups would be a cheap test of a placement.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
static const size_t IDS = 41472000;            // 20 frames of 1920 x 1080 in 8 x 8 tiles
static const size_t BYTES = 6 * IDS * 16;      // six float4 planes
__global__ void k_read(const float4 *p, size_t n, float *out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; s += v.x + v.w; }
    if (s == 12345.f) *out = s;
}
__global__ void k_write(float4 *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void k_rmw(float4 *p, size_t ids, int windows) {   // workgroup b: windows b, b + grid, ... of 4096 ids; planes 0-3 read and written, plane 5 written
    for (int w = 0; w < windows; w++) {
        const size_t base = ((size_t)w * gridDim.x + blockIdx.x) * 4096 % (ids - 4096);
        for (int i = threadIdx.x; i < 4096; i += 256) {
            const size_t id = base + i;
            float4 a = p[id], b = p[ids + id], c = p[2 * ids + id], d = p[3 * ids + id];
            a.x += d.w; b.y += a.z; c.z += b.x; d.w += c.y;
            p[id] = a; p[ids + id] = b; p[2 * ids + id] = c; p[3 * ids + id] = d; p[5 * ids + id] = a;
        }
    }
}
template <class F> static float timed(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int r = 0; r < 3; r++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 3;
}
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8;
    float *out; hipMalloc(&out, 4);
    std::vector<float4 *> bufs;
    for (int i = 0; i < n; i++) { float4 *p = nullptr; if (hipMalloc(&p, BYTES) != hipSuccess) break; hipMemset(p, 0, BYTES); bufs.push_back(p); }
    printf("%zu buffers of %.2f GB\n", bufs.size(), BYTES / 1e9);
    for (int rep = 0; rep < 2; rep++)
        for (size_t i = 0; i < bufs.size(); i++) {
            float4 *p = bufs[i];
            const float tr = timed([&] { hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, p, BYTES / 16, out); });
            const float tw = timed([&] { hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, p, BYTES / 16); });
            const float tm = timed([&] { hipLaunchKernelGGL(k_rmw, dim3(1024), dim3(256), 0, 0, p, IDS, 9); });
            printf("rep %d buffer %zu (%p): read %.0f GB/s  write %.0f GB/s  rmw pattern %.3f ms (%.0f GB/s)\n", rep, i, (void *)p, BYTES / tr / 1e6, BYTES / tw / 1e6, tm,
                   1024.0 * 9 * 4096 * 16 * 9 / tm / 1e6);
        }
    return 0;
}
