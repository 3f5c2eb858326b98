// micro-benchmark of walker step loops: one wave walks a ring inside a 64x64 LDS window
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint8_t u8;
#define WT 64
#define WTP 68
#define ST_FG 1
#define ST_VIS 2
#define TB_RING 0x80
struct Ctx { const u8* img; unsigned long long* out; int li0; unsigned nsteps; };

// ---------------- variant A: branchy C++ (what walker.h compiles to today, events stubbed)
__global__ __launch_bounds__(64) void k_A(Ctx C) {
    __shared__ u8 tile[WT * WTP];
    const int lane = threadIdx.x & 63;
    int noff = 0, dplv = 0; unsigned sel2 = 0;
    if (lane < 8) { const int dx = (int)((0x9224u >> (2 * lane)) & 3u) - 1, dy = (int)((0xA940u >> (2 * lane)) & 3u) - 1; noff = dy * WTP + dx; dplv = dy * 4096 + dx; sel2 = 2u << lane; }
    for (int i = lane; i < WT * WTP; i += 64) tile[i] = C.img[i];
    __syncthreads();
    int li = C.li0; unsigned pl = 1000000;
    unsigned steps = 0, nb = 0, nbatch = 4, steps_b = 0, nofresh = 0, nflush = 0;
    int kopp = 8; int codes = 0, marks = 0; unsigned myS = 0;
    const unsigned s = 77, g2 = C.nsteps;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (true) {
        const int va = li + noff;
        const int v = tile[va];
        const unsigned m_fg = (unsigned)__builtin_amdgcn_ballot_w64((v & ST_FG) != 0), m_vis = (unsigned)__builtin_amdgcn_ballot_w64((v & ST_VIS) != 0);
        const unsigned m_any = m_fg & (0xffu & ~(1u << kopp));
        const unsigned m_un = m_any & ~m_vis;
        const bool fresh = m_un != 0;
        const unsigned m = fresh ? m_un : m_any;
        if (m == 0) break;
        if (fresh && nb) { nb = 0; nflush++; }
        const int k = __builtin_ctz(m);
        if (!fresh && nb == 0) steps_b = steps;
        const unsigned idx = steps & 63u;
        codes = (unsigned)lane == idx ? k : codes;
        if (idx == 63u) { C.out[64 + lane] = codes; }
        steps++;
        const int vk = __builtin_amdgcn_readlane(v, k);
        li += __builtin_amdgcn_readlane(noff, k);
        pl = (unsigned)((int)pl + __builtin_amdgcn_readlane(dplv, k));
        if (fresh) {
            tile[va] = (u8)(v | (int)((sel2 >> k) & 2u));
            marks = (unsigned)lane == idx ? (int)((pl << 4) | (unsigned)((vk | ST_VIS) & 15)) : marks;
        }
        kopp = 7 - k;
        if (pl == s) break;
        if (steps > g2) break;
        if (vk & TB_RING) break;
        if (fresh) { nofresh = 0; nbatch = 4; continue; }
        myS = (unsigned)lane == nb ? ((pl << 3) | (unsigned)k) : myS;
        nb++;
        if (nb == nbatch) { nb = 0; nflush++; nbatch = nbatch * 2 < 64 ? nbatch * 2 : 64; }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { C.out[0] = t1 - t0; C.out[1] = steps; C.out[2] = nflush + steps_b + nofresh; C.out[3] = li; }
    C.out[128 + lane] = (unsigned)codes + (unsigned)marks + myS;
}

// ---------------- variant B: branch-free body, one exit test per step
__global__ __launch_bounds__(64) void k_B(Ctx C) {
    __shared__ u8 tile[WT * WTP];
    const int lane = threadIdx.x & 63;
    int noff = 0, dplv = 0; unsigned sel2 = 0;
    if (lane < 8) { const int dx = (int)((0x9224u >> (2 * lane)) & 3u) - 1, dy = (int)((0xA940u >> (2 * lane)) & 3u) - 1; noff = dy * WTP + dx; dplv = dy * 4096 + dx; sel2 = 2u << lane; }
    for (int i = lane; i < WT * WTP; i += 64) tile[i] = C.img[i];
    __syncthreads();
    int li = C.li0; unsigned pl = 1000000;
    unsigned steps = 0, nb = 0, nbatch = 4, steps_b = 0, nflush = 0;
    unsigned allow = 0xffu;                 // 0xff & ~(1 << kopp)
    int codes = 0, marks = 0; unsigned myS = 0;
    const unsigned s = 77, g2 = C.nsteps;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned stop = 0;
    while (true) {
        const int va = li + noff;
        const int v = tile[va];
        const unsigned m_fg = (unsigned)__builtin_amdgcn_ballot_w64((v & ST_FG) != 0), m_vis = (unsigned)__builtin_amdgcn_ballot_w64((v & ST_VIS) != 0);
        const unsigned m_any = m_fg & allow;
        const unsigned m_un = m_any & ~m_vis;
        const unsigned fr = m_un ? 1u : 0u;
        const unsigned m = m_un ? m_un : m_any;
        // events that must be handled BEFORE this step: dead end, pending states in front of a fresh step
        if (__builtin_expect((m == 0) | (fr & (nb != 0)), 0)) { if (m == 0) break; nb = 0; nflush++; continue; }
        const int k = __builtin_ctz(m);
        steps_b = (fr | nb) ? steps_b : steps;
        const unsigned idx = steps & 63u;
        const bool mine = (unsigned)lane == idx;
        codes = mine ? k : codes;
        steps++;
        const int vk = __builtin_amdgcn_readlane(v, k);
        li += __builtin_amdgcn_readlane(noff, k);
        pl = (unsigned)((int)pl + __builtin_amdgcn_readlane(dplv, k));
        const unsigned visbit = fr ? 2u : 0u;
        tile[va] = (u8)(v | (int)((sel2 >> k) & visbit));
        const unsigned ment = fr ? ((pl << 4) | (unsigned)((vk | ST_VIS) & 15)) : 0u;
        marks = mine ? (int)ment : marks;
        const unsigned psel = fr ? 64u : nb;
        myS = (unsigned)lane == psel ? ((pl << 3) | (unsigned)k) : myS;
        nb += 1u - fr;
        nbatch = fr ? 4u : nbatch;
        allow = 0xffu & ~(0x80u >> k);       // kopp = 7 - k
        stop = (pl == s) | (steps > g2) | ((unsigned)vk >> 7) | (idx == 63u) | (nb == nbatch);
        if (__builtin_expect(stop != 0, 0)) {
            if (pl == s || steps > g2 || (vk & TB_RING)) break;
            if (idx == 63u) C.out[64 + lane] = codes;
            if (nb == nbatch) { nb = 0; nflush++; nbatch = nbatch * 2 < 64 ? nbatch * 2 : 64; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { C.out[0] = t1 - t0; C.out[1] = steps; C.out[2] = nflush + steps_b; C.out[3] = li; }
    C.out[128 + lane] = (unsigned)codes + (unsigned)marks + myS;
}

int main() {
    std::vector<u8> img(WT * WTP, 0);
    // ring: rectangle (6,6)-(57,57), thick corners avoided (plain 4-connected outline => every pixel degree 2, corners too)
    for (int i = 6; i <= 57; i++) { img[6 * WTP + i] = 1; img[57 * WTP + i] = 1; img[i * WTP + 6] = 1; img[i * WTP + 57] = 1; }
    u8* d_img; unsigned long long* d_out; hipMalloc(&d_img, img.size()); hipMalloc(&d_out, 4096);
    hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice);
    Ctx C{d_img, d_out, 6 * WTP + 6, 200000};
    unsigned long long h[4];
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_A, dim3(1), dim3(64), 0, 0, C); hipMemcpy(h, d_out, 32, hipMemcpyDeviceToHost);
        printf("A: %llu cycles, %llu steps -> %.1f cycles/step (chk %llu %llu)\n", h[0], h[1], (double)h[0] / h[1], h[2], h[3]);
        hipLaunchKernelGGL(k_B, dim3(1), dim3(64), 0, 0, C); hipMemcpy(h, d_out, 32, hipMemcpyDeviceToHost);
        printf("B: %llu cycles, %llu steps -> %.1f cycles/step (chk %llu %llu)\n", h[0], h[1], (double)h[0] / h[1], h[2], h[3]);
    }
    return 0;
}
