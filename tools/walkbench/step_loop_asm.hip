// micro-benchmark of the hand-written walker step loop (variant C): one wave walks a ring inside a 64x64 LDS window
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint8_t u8;
#define WT 64
#define WTP 68
// window byte: bit7 FG, bit6 VIS, bit5 RING, bit4 HOME, bit1 JUN, bit0 END
struct Ctx { const u8* img; unsigned long long* out; int li0; unsigned nsteps; };

enum { EV_DEAD = 1, EV_PEND = 2, EV_LIMIT = 3, EV_FLAG = 4, EV_BATCH = 5 };

struct Hot { int li; unsigned pl, allow, steps, limit, nb, nbatch; int vk, k; };

// One asm statement = the step loop; leaves with an event code.  rec: lane (steps & 63) receives ((pl << 3 | k) << 2) | (fresh ? 2 : 0) of every step.
__device__ __forceinline__ int hot_loop(Hot& h, int& rec, int noff, int dplv, unsigned sel, unsigned lds_base) {
    int ev;
    int va, v, vb, w, t2;             // VGPR temporaries
    unsigned m_any, m_un, m, frmask, fr2, fr40, r, tmp, m0save;
    asm volatile(
        "s_mov_b32 %[m0save], m0\n\t"
        "s_and_b32 m0, %[steps], 63\n\t"
        "v_add_u32 %[va], %[li], %[noff]\n\t"
        "ds_read_i8 %[v], %[va]\n\t"
        "L_A%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_gt_i32 vcc, 0, %[v]\n\t"
        "s_and_b32 %[m_any], vcc_lo, %[allow]\n\t"
        "v_cmp_gt_i32 vcc, -64, %[v]\n\t"
        "s_and_b32 %[m_un], %[m_any], vcc_lo\n\t"
        "s_cselect_b32 %[m], %[m_un], %[m_any]\n\t"
        "s_cselect_b32 %[frmask], -1, 0\n\t"
        "s_cmp_eq_u32 %[m], 0\n\t"
        "s_cbranch_scc1 L_dead%=\n\t"
        "s_and_b32 %[tmp], %[frmask], %[nb]\n\t"
        "s_cbranch_scc1 L_pend%=\n\t"
        "s_ff1_i32_b32 %[k], %[m]\n\t"
        "s_and_b32 %[fr40], %[frmask], 64\n\t"
        "v_readlane_b32 %[tmp], %[noff], %[k]\n\t"
        "v_lshrrev_b32 %[t2], %[k], %[sel]\n\t"
        "s_add_i32 %[li], %[li], %[tmp]\n\t"
        "v_and_or_b32 %[t2], %[t2], %[fr40], %[v]\n\t"
        "v_add_u32 %[vb], %[li], %[noff]\n\t"
        "ds_write_b8 %[va], %[t2]\n\t"
        "ds_read_i8 %[w], %[vb]\n\t"
        "v_readlane_b32 %[vk], %[v], %[k]\n\t"
        "v_readlane_b32 %[tmp], %[dplv], %[k]\n\t"
        "s_and_b32 %[fr2], %[frmask], 2\n\t"
        "s_add_i32 %[pl], %[pl], %[tmp]\n\t"
        "s_lshl3_add_u32 %[r], %[pl], %[k]\n\t"
        "s_lshl2_add_u32 %[r], %[r], %[fr2]\n\t"
        "v_writelane_b32 %[rec], %[r], m0\n\t"
        "s_add_i32 %[nb], %[nb], 1\n\t"
        "s_andn2_b32 %[nb], %[nb], %[frmask]\n\t"
        "s_lshr_b32 %[tmp], 0x80, %[k]\n\t"
        "s_andn2_b32 %[allow], 0xff, %[tmp]\n\t"
        "s_add_i32 %[steps], %[steps], 1\n\t"
        "s_add_i32 m0, m0, 1\n\t"
        "s_and_b32 %[tmp], %[vk], 0x30\n\t"
        "s_cbranch_scc1 L_flag%=\n\t"
        "s_cmp_ge_u32 %[steps], %[limit]\n\t"
        "s_cbranch_scc1 L_limit%=\n\t"
        "s_cmp_eq_u32 %[nb], %[nbatch]\n\t"
        "s_cbranch_scc1 L_batch%=\n\t"
        "L_B%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_gt_i32 vcc, 0, %[w]\n\t"
        "s_and_b32 %[m_any], vcc_lo, %[allow]\n\t"
        "v_cmp_gt_i32 vcc, -64, %[w]\n\t"
        "s_and_b32 %[m_un], %[m_any], vcc_lo\n\t"
        "s_cselect_b32 %[m], %[m_un], %[m_any]\n\t"
        "s_cselect_b32 %[frmask], -1, 0\n\t"
        "s_cmp_eq_u32 %[m], 0\n\t"
        "s_cbranch_scc1 L_dead%=\n\t"
        "s_and_b32 %[tmp], %[frmask], %[nb]\n\t"
        "s_cbranch_scc1 L_pend%=\n\t"
        "s_ff1_i32_b32 %[k], %[m]\n\t"
        "s_and_b32 %[fr40], %[frmask], 64\n\t"
        "v_readlane_b32 %[tmp], %[noff], %[k]\n\t"
        "v_lshrrev_b32 %[t2], %[k], %[sel]\n\t"
        "s_add_i32 %[li], %[li], %[tmp]\n\t"
        "v_and_or_b32 %[t2], %[t2], %[fr40], %[w]\n\t"
        "v_add_u32 %[va], %[li], %[noff]\n\t"
        "ds_write_b8 %[vb], %[t2]\n\t"
        "ds_read_i8 %[v], %[va]\n\t"
        "v_readlane_b32 %[vk], %[w], %[k]\n\t"
        "v_readlane_b32 %[tmp], %[dplv], %[k]\n\t"
        "s_and_b32 %[fr2], %[frmask], 2\n\t"
        "s_add_i32 %[pl], %[pl], %[tmp]\n\t"
        "s_lshl3_add_u32 %[r], %[pl], %[k]\n\t"
        "s_lshl2_add_u32 %[r], %[r], %[fr2]\n\t"
        "v_writelane_b32 %[rec], %[r], m0\n\t"
        "s_add_i32 %[nb], %[nb], 1\n\t"
        "s_andn2_b32 %[nb], %[nb], %[frmask]\n\t"
        "s_lshr_b32 %[tmp], 0x80, %[k]\n\t"
        "s_andn2_b32 %[allow], 0xff, %[tmp]\n\t"
        "s_add_i32 %[steps], %[steps], 1\n\t"
        "s_add_i32 m0, m0, 1\n\t"
        "s_and_b32 %[tmp], %[vk], 0x30\n\t"
        "s_cbranch_scc1 L_flag%=\n\t"
        "s_cmp_ge_u32 %[steps], %[limit]\n\t"
        "s_cbranch_scc1 L_limit%=\n\t"
        "s_cmp_eq_u32 %[nb], %[nbatch]\n\t"
        "s_cbranch_scc1 L_batch%=\n\t"
        "s_branch L_A%=\n\t"
        "L_batch%=:\n\t"
        "s_mov_b32 %[ev], 5\n\t"
        "s_branch L_out%=\n\t"
        "L_dead%=:\n\t"
        "s_mov_b32 %[ev], 1\n\t"
        "s_branch L_out%=\n\t"
        "L_pend%=:\n\t"
        "s_mov_b32 %[ev], 2\n\t"
        "s_branch L_out%=\n\t"
        "L_flag%=:\n\t"
        "s_mov_b32 %[ev], 4\n\t"
        "s_branch L_out%=\n\t"
        "L_limit%=:\n\t"
        "s_mov_b32 %[ev], 3\n\t"
        "L_out%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b32 m0, %[m0save]\n\t"
        : [ev] "=&s"(ev), [li] "+s"(h.li), [pl] "+s"(h.pl), [allow] "+s"(h.allow), [steps] "+s"(h.steps), [nb] "+s"(h.nb), [vk] "+s"(h.vk), [k] "+s"(h.k),
          [rec] "+v"(rec), [va] "=&v"(va), [v] "=&v"(v), [vb] "=&v"(vb), [w] "=&v"(w), [t2] "=&v"(t2),
          [m_any] "=&s"(m_any), [m_un] "=&s"(m_un), [m] "=&s"(m), [frmask] "=&s"(frmask), [fr2] "=&s"(fr2), [fr40] "=&s"(fr40), [r] "=&s"(r), [tmp] "=&s"(tmp),
          [m0save] "=&s"(m0save)
        : [limit] "s"(h.limit), [nbatch] "s"(h.nbatch), [noff] "v"(noff), [dplv] "v"(dplv), [sel] "v"(sel), "s"(lds_base)
        : "vcc", "scc", "memory");
    return ev;
}

__global__ __launch_bounds__(64) void k_C(Ctx C) {
    __shared__ u8 tile[WT * WTP];
    const int lane = threadIdx.x & 63;
    int noff = 0, dplv = 0; unsigned sel = 0;
    if (lane < 8) { const int dx = (int)((0x9224u >> (2 * lane)) & 3u) - 1, dy = (int)((0xA940u >> (2 * lane)) & 3u) - 1; noff = dy * WTP + dx; dplv = dy * 4096 + dx; sel = 0x40u << lane; }
    for (int i = lane; i < WT * WTP; i += 64) tile[i] = C.img[i];
    __syncthreads();
    const unsigned lds_base = (unsigned)(uintptr_t)tile;
    Hot h; h.li = __builtin_amdgcn_readfirstlane(C.li0 + (int)lds_base); h.pl = 1000000; h.allow = 0xff; h.steps = 0; h.nb = 0; h.nbatch = 4; h.vk = 0; h.k = 0;
    int rec = 0; unsigned nflush = 0, nev = 0; unsigned long long chk = 0;
    const unsigned g2 = C.nsteps;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (true) {
        unsigned lim = (h.steps & ~63u) + 64u; if (lim > g2 + 1) lim = g2 + 1;
        h.limit = lim;
        const int ev = hot_loop(h, rec, noff, dplv, sel, lds_base);
        nev++;
        if (ev == EV_DEAD) break;
        if (ev == EV_PEND) { h.nb = 0; h.nbatch = 4; nflush++; continue; }
        if (ev == EV_BATCH) { h.nb = 0; nflush++; h.nbatch = h.nbatch * 2 < 64 ? h.nbatch * 2 : 64; continue; }
        if (ev == EV_FLAG) break;
        if (ev == EV_LIMIT) { if (h.steps > g2) break; if ((h.steps & 63u) == 0) { chk += (unsigned)rec; C.out[64 + lane] = rec; } continue; }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { C.out[0] = t1 - t0; C.out[1] = h.steps; C.out[2] = nflush; C.out[3] = h.li; C.out[4] = nev; C.out[5] = h.pl; }
    C.out[128 + lane] = (unsigned)rec + chk;
}

int main() {
    std::vector<u8> img(WT * WTP, 0);
    for (int i = 6; i <= 57; i++) { img[6 * WTP + i] = 0x80; img[57 * WTP + i] = 0x80; img[i * WTP + 6] = 0x80; img[i * WTP + 57] = 0x80; }
    u8* d_img; unsigned long long* d_out; (void)hipMalloc(&d_img, img.size()); (void)hipMalloc(&d_out, 4096);
    (void)hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice);
    Ctx C{d_img, d_out, 6 * WTP + 6, 200000};
    unsigned long long h[6];
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_C, dim3(1), dim3(64), 0, 0, C); (void)hipMemcpy(h, d_out, 48, hipMemcpyDeviceToHost);
        printf("D: %llu cycles, %llu steps -> %.1f cycles/step (flushes %llu li %llu events %llu pl %llu)\n", h[0], h[1], (double)h[0] / h[1], h[2], h[3], h[4], h[5]);
    }
    return 0;
}
