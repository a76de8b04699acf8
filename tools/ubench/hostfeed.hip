// Probe (round 6): can a RUNNING kernel and the host exchange work through host-coherent memory on this box?
//   1. system-scope compare-and-swap from the device on hipHostMalloc'ed coherent memory while the host does locked
//      compare-and-swaps on the same word: both sides increment a shared counter N times; lost updates = atomics are not atomic across PCIe.
//   2. latency of a system-scope load of a host word from the device (what a workgroup pays to look for appended frames).
//   3. a persistent kernel that the host keeps feeding: the device closes the feed with a CAS when it runs dry; the host's append fails from then on.
// Build: hipcc -O2 --offload-arch=gfx950 -o hostfeed hostfeed.hip ; run: ./hostfeed
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_inc(unsigned *w, int n, unsigned long long *clk) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
        unsigned old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        while (!__hip_atomic_compare_exchange_strong(w, &old, old + 1u, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {}
    }
    if (clk) *clk = __builtin_amdgcn_s_memtime() - t0;
}

__global__ void k_load(const unsigned *w, int n, unsigned long long *out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
    for (int i = 0; i < n; i++) s += __hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    out[0] = __builtin_amdgcn_s_memtime() - t0;
    out[1] = s;
}

// feed word: frames published by the host in bits 0..30, bit 31 = closed (set by the device).  The kernel "renders" a frame by spinning `work` clocks.
__global__ void k_feed(unsigned *feed, unsigned *consumed_out, int work) {
    unsigned done = 0;
    for (;;) {
        unsigned v = __hip_atomic_load(feed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((v & 0x7fffffffu) > done) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)work) {}
            done++;
            continue;
        }
        if (__hip_atomic_compare_exchange_strong(feed, &v, v | 0x80000000u, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;  // dry: close
    }
    *consumed_out = done;
}

int main() {
    unsigned *hw = nullptr, *dw = nullptr;
    CK(hipHostMalloc((void **)&hw, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&dw, hw, 0));
    unsigned long long *clk = nullptr;
    CK(hipMalloc((void **)&clk, 64));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // 1. contended increments
    const int N = 200000;
    hw[0] = 0;
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, s, dw, N, clk);
    auto *aw = reinterpret_cast<std::atomic<unsigned> *>(hw);
    for (int i = 0; i < N; i++) { unsigned old = aw->load(); while (!aw->compare_exchange_weak(old, old + 1u)) {} }
    CK(hipStreamSynchronize(s));
    unsigned long long c = 0;
    CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    std::printf("1. contended CAS increments: host %d + device %d -> %u (%s); device %.2f us per CAS (100 MHz counter)\n", N, N, hw[0], hw[0] == 2u * N ? "ATOMIC" : "LOST UPDATES", c / 100.0 / N);
    // 2. load latency
    hipLaunchKernelGGL(k_load, dim3(1), dim3(1), 0, s, dw, 10000, clk);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    std::printf("2. system-scope load of a host word from the device: %.2f us each\n", c / 100.0 / 10000);
    // 3. feed protocol, many rounds: the host appends as fast as it can for a while, then stops; frames accepted by the host's CAS must equal frames consumed
    unsigned *cons = nullptr;
    CK(hipHostMalloc((void **)&cons, 64, hipHostMallocMapped | hipHostMallocCoherent));
    unsigned *dcons = nullptr;
    CK(hipHostGetDevicePointer((void **)&dcons, cons, 0));
    int bad = 0; unsigned long long total = 0;
    for (int round = 0; round < 300; round++) {
        hw[0] = 1;  // one frame to start with
        cons[0] = 0xffffffffu;
        hipLaunchKernelGGL(k_feed, dim3(1), dim3(1), 0, s, dw, dcons, 200 + 37 * (round % 13));
        unsigned accepted = 1;
        const int want = 1 + (round * 7) % 200;
        for (int i = 0; i < want; i++) {
            unsigned exp = accepted;
            if (!aw->compare_exchange_strong(exp, accepted + 1u)) break;  // closed
            accepted++;
            if ((round & 3) == 0) for (volatile int d = 0; d < (round % 50) * 20; d++) {}
        }
        CK(hipStreamSynchronize(s));
        if (cons[0] != accepted) { bad++; if (bad < 5) std::printf("   round %d: accepted %u consumed %u word %08x\n", round, accepted, cons[0], hw[0]); }
        total += accepted;
    }
    std::printf("3. feed protocol: 300 rounds, %llu frames, mismatching rounds: %d (%s)\n", total, bad, bad ? "BROKEN" : "OK");
    // 4. same with the host's appends racing the close on purpose (tiny work per frame)
    bad = 0; total = 0;
    for (int round = 0; round < 2000; round++) {
        hw[0] = 1; cons[0] = 0xffffffffu;
        hipLaunchKernelGGL(k_feed, dim3(1), dim3(1), 0, s, dw, dcons, 20);
        unsigned accepted = 1;
        for (;;) { unsigned exp = accepted; if (!aw->compare_exchange_strong(exp, accepted + 1u)) break; accepted++; if (accepted > 100000) break; }
        if (accepted > 100000) std::printf("   round %d never closed\n", round);
        CK(hipStreamSynchronize(s));
        if (cons[0] != accepted && accepted <= 100000) bad++;
        total += accepted;
    }
    std::printf("4. racing close: 2000 rounds, %llu frames, mismatching rounds: %d (%s)\n", total, bad, bad ? "BROKEN" : "OK");
    return 0;
}
