// Map of the streaming WRITE bandwidth over a large allocation, window by window (tools/ubench/placement.hip: 4 GB buffers from hipMalloc differ by 25 % in write bandwidth
// and by nothing in read bandwidth; the renderer's timing regime follows its path-state buffer).   writemap [GiB total] [MiB per window]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_write(float4 *p, size_t n, int reps) {
    for (int r = 0; r < reps; r++)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, (float)r);
}
__global__ void k_read(const float4 *p, size_t n, int reps, float *out) {
    float s = 0.f;
    for (int r = 0; r < reps; r++)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; s += v.x + v.w; }
    if (s == 12345.f) *out = s;
}
int main(int argc, char **argv) {
    const size_t total = (size_t)(argc > 1 ? atoi(argv[1]) : 48) << 30, win = (size_t)(argc > 2 ? atoi(argv[2]) : 1024) << 20;
    char *p = nullptr; float *out; hipMalloc(&out, 4);
    if (hipMalloc(&p, total) != hipSuccess) { printf("allocation failed\n"); return 2; }
    hipMemset(p, 0, total); hipDeviceSynchronize();
    const int reps = (int)(((size_t)8 << 30) / win) > 0 ? (int)(((size_t)8 << 30) / win) : 1;   // 8 GiB moved per measurement
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    printf("%zu GiB at %p, windows of %zu MiB; write GB/s (read GB/s) per window:\n", total >> 30, (void *)p, win >> 20);
    for (size_t off = 0; off + win <= total; off += win) {
        float mw, mr;
        hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, (float4 *)(p + off), win / 16, 1);
        hipEventRecord(a); hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, (float4 *)(p + off), win / 16, reps); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&mw, a, b);
        hipEventRecord(a); hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, (const float4 *)(p + off), win / 16, reps, out); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&mr, a, b);
        printf("%5.0f(%4.0f)%s", (double)win * reps / mw / 1e6, (double)win * reps / mr / 1e6, ((off / win) % 8 == 7) ? "\n" : " ");
    }
    printf("\n");
    return 0;
}
