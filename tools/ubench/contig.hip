// hipMalloc against hipExtMallocWithFlags(hipDeviceMallocContiguous), 4 GiB each: streaming read bandwidth over the whole buffer and over WINDOWS of it (all CUs reading
// one window of W bytes again and again: an access pattern that is local in the address space, like a wavefront renderer's moving window of path state), and a
// dependent gather.  Question (profiles/r04_context_regimes.txt): the render kernel is 17-30 % slower on physically contiguous buffers although a full-range stream is not --
// is a window of a contiguous range served by fewer HBM channels than a window of scattered pages?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read(const float4 *p, size_t n, int reps, float *out) {
    float s = 0.f;
    for (int r = 0; r < reps; r++)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.f) *out = s;
}
static const size_t BYTES = (size_t)4 << 30;
static float bw(const char *base, size_t off, size_t w, float *out) {  // GB/s reading [off, off + w) so often that 8 GiB move
    const int reps = (int)(((size_t)8 << 30) / w);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, (const float4 *)(base + off), w / 16, 1, out);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, (const float4 *)(base + off), w / 16, reps, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return (float)((double)w * reps / ms / 1e6);
}
int main() {
    float *out; hipMalloc(&out, 4);
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, BYTES) != hipSuccess) return 2;
    const hipError_t e = hipExtMallocWithFlags(&b, BYTES, hipDeviceMallocContiguous);
    if (e != hipSuccess) { printf("contiguous allocation refused: %s\n", hipGetErrorString(e)); return 0; }
    hipMemset(a, 0, BYTES); hipMemset(b, 0, BYTES);
    printf("window        offsets (GiB)         hipMalloc GB/s                      contiguous GB/s\n");
    for (size_t w : {BYTES, (size_t)1 << 30, (size_t)512 << 20, (size_t)256 << 20}) {   // windows larger than the 256 MiB Infinity Cache, so that HBM serves them
        printf("%5zu MiB   ", w >> 20);
        float ra[4], rb[4]; int k = 0;
        for (size_t off = 0; off + w <= BYTES && k < 4; off += (BYTES - w) / 3 ? (BYTES - w) / 3 : BYTES, k++) { ra[k] = bw((const char *)a, off, w, out); rb[k] = bw((const char *)b, off, w, out); }
        printf("%d windows   ", k);
        for (int i = 0; i < k; i++) printf("%6.0f", ra[i]);
        printf("      |  ");
        for (int i = 0; i < k; i++) printf("%6.0f", rb[i]);
        printf("\n");
    }
    return 0;
}
