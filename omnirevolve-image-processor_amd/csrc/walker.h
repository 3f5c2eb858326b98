// csrc/walker.h -- the centerline walker of stage 04 (04_find_contours.py trace_centerlines, 04:137-205).
// Shared verbatim by the HIP kernel k_walk (raster04.hip) and by tests/host/walk_harness.cpp, which compiles
// this header with g++ to unit-test the serial walk logic on the CPU (test infrastructure; not a product path).
#pragma once
#include <cstdint>
#include "../../include/orip.h"
#if defined(__HIPCC__)
#define ORIP_HD __host__ __device__
#else
#define ORIP_HD
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define ORIP_FETCH_INC(p) atomicAdd((p), 1u)
#else
#define ORIP_FETCH_INC(p) ((*(p))++)
#endif
typedef uint8_t u8;
#define ST_FG 1
#define ST_VIS 2
#define ST_END 4
#define ST_JUN 8

struct WalkArgs {
    int H, W; int64_t plane;
    u8* st;                            // [K,H,W]
    const unsigned* keys; const unsigned* lin; const unsigned* comp_start; unsigned nc;
    long long total_fg[ORIP_MAX_LAYERS];
    // count pass outputs / write pass inputs
    unsigned long long* comp_pts; unsigned* comp_paths;            // per component (kept paths only)
    const unsigned long long* pts_base; const unsigned* path_base; // exclusive scans over components (global)
    unsigned long long layer_pts_base[ORIP_MAX_LAYERS]; unsigned layer_path_base[ORIP_MAX_LAYERS];
    int32_t* pts[ORIP_MAX_LAYERS]; int64_t* off[ORIP_MAX_LAYERS];
    // cycle expansion descriptors: (layer, dst point index, period, count)
    unsigned long long* desc; unsigned* n_desc; unsigned desc_cap;
};


template <bool WRITE>
ORIP_HD inline void walk_component(const WalkArgs& A, unsigned c) {
    const int NBX[8] = {-1, 0, 1, -1, 1, -1, 0, 1};   // NEIGH8 (dx,dy), 04:12
    const int NBY[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
    const unsigned b = A.comp_start[c], e = A.comp_start[c + 1];
    const int layer = (int)(A.keys[b] >> 26);
    u8* st = A.st + A.plane * layer;
    const int W = A.W, H = A.H;
    const long long fg_comp = (long long)(e - b), total_fg = A.total_fg[layer];
    unsigned long long n_pts = 0; unsigned n_paths = 0;
    int32_t* out = nullptr; int64_t* off = nullptr; unsigned long long wpos = 0, wend = 0; unsigned wpath = 0;
    if (WRITE) {
        out = A.pts[layer]; off = A.off[layer];
        wpos = A.pts_base[c] - A.layer_pts_base[layer];
        wpath = A.path_base[c] - A.layer_path_base[layer];
        wend = A.pts_base[c + 1] - A.layer_pts_base[layer];
    }
    // Paths shorter than 5 points are dropped after the fact, so their points must never land outside this component's own
    // output range [wpos0, wend) (the range holds exactly the kept paths; anything inside it is overwritten by a later kept path).
    auto emit = [&](unsigned long long pos, int x, int y) { if (WRITE && pos < wend) { out[2 * pos] = x; out[2 * pos + 1] = y; } };
    // ---- phase 1: walks from endpoints (04:144-171)
    for (unsigned q = b; q < e; q++) {
        unsigned s = A.lin[q];
        u8 sv = st[s];
        if (!(sv & ST_END) || (sv & ST_VIS)) continue;
        int px = (int)(s % W), py = (int)(s / W), pvx = -1, pvy = -1;
        unsigned long long len = 1; emit(wpos, px, py);
        st[s] = sv | ST_VIS;
        long long guard = 0;
        while (true) {
            int nx = -1, ny = -1;
            for (int k = 0; k < 8; k++) {
                int xx = px + NBX[k], yy = py + NBY[k];
                if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                u8 v = st[(size_t)yy * W + xx];
                if (!(v & ST_FG) || (v & ST_VIS)) continue;
                if (xx == pvx && yy == pvy) continue;
                nx = xx; ny = yy; break;
            }
            if (nx < 0) break;
            emit(wpos + len, nx, ny); len++;
            size_t j = (size_t)ny * W + nx;
            u8 v = st[j]; st[j] = v | ST_VIS;
            pvx = px; pvy = py; px = nx; py = ny;
            if (v & (ST_JUN | ST_END)) break;
            guard++;
            if (guard > total_fg * 2) break;
        }
        if (len >= 5) {   // >=2 to be a path (04:168) and >=5 to survive vectorize_layer (04:224)
            n_pts += len; n_paths++;
            if (WRITE) { wpos += len; off[wpath + 1] = (int64_t)wpos; wpath++; }
        }
    }
    // ---- phase 2: leftovers / cycles (04:174-205)
    for (unsigned q = b; q < e; q++) {
        unsigned s = A.lin[q];
        u8 sv = st[s];
        if (sv & ST_VIS) continue;
        const int x0 = (int)(s % W), y0 = (int)(s / W);
        int px = x0, py = y0, pvx = -1, pvy = -1;
        unsigned long long len = 1; emit(wpos, px, py);
        st[s] = sv | ST_VIS;
        long long guard = 0;
        // Brent cycle detection on the (prev,cur) state; reset whenever a fresh pixel is consumed
        int tpx = -2, tpy = -2, tcx = -2, tcy = -2; long long power = 1, lam = 0;
        bool closed_on_start = false, expanded = false;
        while (true) {
            int nx = -1, ny = -1, ax = -1, ay = -1;
            for (int k = 0; k < 8; k++) {
                int xx = px + NBX[k], yy = py + NBY[k];
                if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                u8 v = st[(size_t)yy * W + xx];
                if (!(v & ST_FG)) continue;
                if (xx == pvx && yy == pvy) continue;
                if (ax < 0) { ax = xx; ay = yy; }
                if (!(v & ST_VIS)) { nx = xx; ny = yy; break; }
            }
            bool fresh = nx >= 0;
            if (!fresh) { nx = ax; ny = ay; }
            if (nx < 0) break;
            emit(wpos + len, nx, ny); len++;
            if (fresh) { size_t j = (size_t)ny * W + nx; st[j] |= ST_VIS; }
            pvx = px; pvy = py; px = nx; py = ny;
            if (px == x0 && py == y0) { closed_on_start = true; break; }
            guard++;
            if (guard > fg_comp * 4) break;
            if (fresh) { tpx = pvx; tpy = pvy; tcx = px; tcy = py; power = 1; lam = 0; }
            else {
                lam++;
                if (pvx == tpx && pvy == tpy && px == tcx && py == tcy) {
                    // the last `lam` steps repeat forever (no fresh pixel is reachable from the cycle and the start is not on it):
                    // remaining steps until the guard fires
                    long long remaining = fg_comp * 4 + 1 - guard;
                    if (remaining > 0) {
                        if (WRITE) {
                            unsigned slot = ORIP_FETCH_INC(A.n_desc);
                            if (slot < A.desc_cap) {
                                unsigned long long* d = A.desc + 4ull * slot;
                                d[0] = (unsigned long long)layer; d[1] = wpos + len; d[2] = (unsigned long long)lam; d[3] = (unsigned long long)remaining;
                            }
                        }
                        // position after the remaining steps = cycle point (remaining mod lam) steps ahead; needed for the closing test
                        long long adv = remaining % lam;
                        if (adv) {
                            // replay adv steps (no fresh pixels by construction)
                            for (long long t = 0; t < adv; t++) {
                                int bx = -1, by = -1;
                                for (int k = 0; k < 8; k++) {
                                    int xx = px + NBX[k], yy = py + NBY[k];
                                    if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                                    if (!(st[(size_t)yy * W + xx] & ST_FG)) continue;
                                    if (xx == pvx && yy == pvy) continue;
                                    bx = xx; by = yy; break;
                                }
                                pvx = px; pvy = py; px = bx; py = by;
                            }
                        }
                        len += (unsigned long long)remaining;
                    }
                    expanded = true;
                    break;
                }
                if (lam == power) { tpx = pvx; tpy = pvy; tcx = px; tcy = py; power <<= 1; lam = 0; }
            }
        }
        (void)closed_on_start; (void)expanded;
        if (len >= 2) {
            int ddx = x0 - px, ddy = y0 - py;
            if (ddx * ddx + ddy * ddy < 3) {   // hypot < 1.5 on integers  <=>  d2 in {0,1,2}
                emit(wpos + len, x0, y0); len++;
            }
            if (len >= 5) {
                n_pts += len; n_paths++;
                if (WRITE) { wpos += len; off[wpath + 1] = (int64_t)wpos; wpath++; }
            }
        }
    }
    if (!WRITE) { A.comp_pts[c] = n_pts; A.comp_paths[c] = n_paths; }
}

