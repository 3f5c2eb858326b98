// Dependent-load latency seen by ONE wave (development aid): a pointer chase over a random cycle of 64-byte lines in buffers of growing size,
// with vector loads (global_load_dword, all lanes the same address) and with scalar loads (s_load_dword).  Pass 1 is cold for this CU's XCD
// (the buffer was written by a kernel spread over the chip), pass 2 repeats the same chase.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <numeric>

__global__ void k_fill(unsigned* buf, const unsigned* next, size_t nlines) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nlines) buf[i * 16] = next[i] * 16;
}
__global__ __launch_bounds__(64) void k_chase_v(const unsigned* __restrict__ buf, int steps, unsigned long long* out) {
    unsigned p = 0;
    for (int pass = 0; pass < 2; pass++) {
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < steps; i++) p = __builtin_nontemporal_load(buf + p) * 0 + buf[p];
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) out[pass] = t1 - t0;
    }
    if (threadIdx.x == 0) out[4] = p;
}
__global__ __launch_bounds__(64) void k_chase_v1(const unsigned* __restrict__ buf, int steps, unsigned long long* out) {
    unsigned p = 0;
    for (int pass = 0; pass < 2; pass++) {
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < steps; i++) p = buf[p];
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) out[pass] = t1 - t0;
    }
    if (threadIdx.x == 0) out[4] = p;
}
__global__ __launch_bounds__(64) void k_chase_s(const unsigned* __restrict__ buf, int steps, unsigned long long* out) {
    unsigned p = 0;
    for (int pass = 0; pass < 2; pass++) {
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < steps; i++) p = __builtin_amdgcn_readfirstlane(buf[__builtin_amdgcn_readfirstlane(p)]);
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) out[pass] = t1 - t0;
    }
    if (threadIdx.x == 0) out[4] = p;
}
// sweep first (as k_greedy_nn_fast does), then chase
__global__ __launch_bounds__(64) void k_sweep_chase(const unsigned* __restrict__ buf, size_t ndw, int steps, unsigned long long* out) {
    unsigned acc = 0;
    const uint4* b4 = reinterpret_cast<const uint4*>(buf);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (size_t t = threadIdx.x; t < ndw / 4; t += 64) { const uint4 v = b4[t]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned p = acc == 0x12345u ? 16u : 0u;
    for (int i = 0; i < steps; i++) p = buf[p];
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[4] = p; }
}

int main() {
    unsigned long long* d; (void)hipMalloc(&d, 64); unsigned long long h[8];
    std::mt19937 rng(1);
    for (size_t kb : {64, 512, 640, 2048, 16384, 262144}) {
        const size_t nlines = kb * 1024 / 64;
        std::vector<unsigned> perm(nlines), next(nlines);
        std::iota(perm.begin(), perm.end(), 0u);
        std::shuffle(perm.begin() + 1, perm.end(), rng);
        for (size_t i = 0; i < nlines; i++) next[perm[i]] = perm[(i + 1) % nlines];
        unsigned *buf, *dn; (void)hipMalloc(&buf, nlines * 64); (void)hipMalloc(&dn, nlines * 4);
        (void)hipMemcpy(dn, next.data(), nlines * 4, hipMemcpyHostToDevice);
        const int steps = (int)std::min<size_t>(nlines, 8192);
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, 0, buf, dn, nlines);
        hipLaunchKernelGGL(k_chase_v1, dim3(1), dim3(64), 0, 0, buf, steps, d); (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%8zu KB  vector  cold %7.1f  again %7.1f cycles/load", kb, (double)h[0] / steps, (double)h[1] / steps);
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, 0, buf, dn, nlines);
        hipLaunchKernelGGL(k_chase_s, dim3(1), dim3(64), 0, 0, buf, steps, d); (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("   scalar  cold %7.1f  again %7.1f", (double)h[0] / steps, (double)h[1] / steps);
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, 0, buf, dn, nlines);
        hipLaunchKernelGGL(k_sweep_chase, dim3(1), dim3(64), 0, 0, buf, nlines * 16, steps, d); (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("   sweep %9.0f cycles, then vector %7.1f\n", (double)h[0], (double)h[1] / steps);
        (void)hipFree(buf); (void)hipFree(dn);
    }
    // clock: s_memtime ticks per microsecond
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned* buf; (void)hipMalloc(&buf, 64 * 1024); (void)hipMemset(buf, 0, 64 * 1024);
    hipEventRecord(e0); hipLaunchKernelGGL(k_chase_v1, dim3(1), dim3(64), 0, 0, buf, 200000, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("s_memtime: %.1f ticks per microsecond (%.3f ms for %llu ticks)\n", (double)(h[0] + h[1]) / (ms * 1e3), ms, h[0] + h[1]);
    return 0;
}
