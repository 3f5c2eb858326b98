// csrc/raster03.hip -- stage 03 (03_edge_detect.py process_color, 03:13-40) on gfx950:
//   open/close with MORPH_ELLIPSE(k,k)   -> orip_morph_open_close (raster02.hip)
//   k_blur_sobel_nms : cv2.GaussianBlur((k,k),0) + Canny front half (Sobel 3x3, L1 magnitude, NMS) fused in LDS
//   hysteresis       : Canny back half as 8-connected components of candidate pixels (union-find CCL),
//                      a component is an edge iff it holds a strong pixel (order independent)
// Also hosts the generic union-find CCL kernels reused by stage 04.
#include "orip_ctx.h"
#include <algorithm>

int orip_morph_open_close(orip_ctx* c, const u8* src, u8* dst, int K, int shape, int k, int open_iters, int close_iters, bool labels_mode, bool unpack = true);

// ------------------------------------------------------------------------------------------------
// Gaussian (fixed tables) + Sobel + NMS.  Output map: 0 weak candidate, 1 not an edge,
// 2 strong.  Tile 64x16 outputs; LDS: mask tile (halo r+2), blurred tile (halo 2), magnitude (halo 1).
// ------------------------------------------------------------------------------------------------
#define ET_X 64
#define ET_Y 16
#define EH_MAX 5   // r(<=3) + 2

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) { p = p < 0 ? -p : 2 * (n - 1) - p; }
    return p;
}

__global__ __launch_bounds__(256) void k_blur_sobel_nms(const u8* __restrict__ masks, u8* __restrict__ map, int H, int W, int gk, int low, int high,
                                                        unsigned long long* __restrict__ cand_bits, unsigned long long* __restrict__ strong_bits, int Ww) {
    __shared__ u8 M[ET_Y + 2 * EH_MAX][ET_X + 2 * EH_MAX];
    __shared__ u8 B[ET_Y + 4][ET_X + 4];
    __shared__ short MAG[ET_Y + 2][ET_X + 2];
    const int r = gk >> 1, hm = r + 2;
    const size_t plane = (size_t)H * W;
    const u8* src = masks + plane * blockIdx.z;
    u8* dst = map + plane * blockIdx.z;
    const int x0 = blockIdx.x * ET_X, y0 = blockIdx.y * ET_Y;
    const int mw = ET_X + 2 * hm, mh = ET_Y + 2 * hm;
    for (int i = threadIdx.x; i < mw * mh; i += blockDim.x) {
        int ty = i / mw, tx = i % mw;
        int y = reflect101(y0 + ty - hm, H), x = reflect101(x0 + tx - hm, W);
        M[ty][tx] = src[(size_t)y * W + x];
    }
    __syncthreads();
    // blurred tile with halo 2; positions outside the image take the value at the clamped position (BORDER_REPLICATE for Sobel)
    int w1, w2, w3, w0, shift;   // symmetric weights: w0 centre
    if (gk == 3) { w0 = 2; w1 = 1; w2 = 0; w3 = 0; shift = 4; }
    else if (gk == 5) { w0 = 6; w1 = 4; w2 = 1; w3 = 0; shift = 8; }
    else { w0 = 72; w1 = 56; w2 = 28; w3 = 8; shift = 16; }
    const int wt[4] = {w0, w1, w2, w3};
    for (int i = threadIdx.x; i < (ET_Y + 4) * (ET_X + 4); i += blockDim.x) {
        int ty = i / (ET_X + 4), tx = i % (ET_X + 4);
        int y = y0 + ty - 2, x = x0 + tx - 2;
        int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
        // tile coordinates of the clamped centre (always inside the staged M because |y-yc|<=2 only at image borders)
        int my = yc - (y0 - hm), mx = xc - (x0 - hm);
        long long s = 0;
        for (int a = -r; a <= r; a++) {
            int rowsum = 0;
            for (int b = -r; b <= r; b++) rowsum += wt[abs(b)] * (int)M[my + a][mx + b];
            s += (long long)wt[abs(a)] * rowsum;
        }
        B[ty][tx] = (u8)((s + (1LL << (shift - 1))) >> shift);
    }
    __syncthreads();
    // magnitude with halo 1 (0 outside the image)
    for (int i = threadIdx.x; i < (ET_Y + 2) * (ET_X + 2); i += blockDim.x) {
        int ty = i / (ET_X + 2), tx = i % (ET_X + 2);
        int y = y0 + ty - 1, x = x0 + tx - 1;
        int m = 0;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            int by = ty + 1, bx = tx + 1;
            int gx = ((int)B[by - 1][bx + 1] + 2 * (int)B[by][bx + 1] + (int)B[by + 1][bx + 1]) - ((int)B[by - 1][bx - 1] + 2 * (int)B[by][bx - 1] + (int)B[by + 1][bx - 1]);
            int gy = ((int)B[by + 1][bx - 1] + 2 * (int)B[by + 1][bx] + (int)B[by + 1][bx + 1]) - ((int)B[by - 1][bx - 1] + 2 * (int)B[by - 1][bx] + (int)B[by - 1][bx + 1]);
            m = abs(gx) + abs(gy);
        }
        MAG[ty][tx] = (short)m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ET_Y * ET_X; i += blockDim.x) {       // a wave = one row of the tile (ET_X == 64)
        int ty = i / ET_X, tx = i % ET_X;
        int y = y0 + ty, x = x0 + tx;
        const bool in = y < H && x < W;
        u8 res = 1;
        if (in) {
            int by = ty + 2, bx = tx + 2;
            int xs = ((int)B[by - 1][bx + 1] + 2 * (int)B[by][bx + 1] + (int)B[by + 1][bx + 1]) - ((int)B[by - 1][bx - 1] + 2 * (int)B[by][bx - 1] + (int)B[by + 1][bx - 1]);
            int ys = ((int)B[by + 1][bx - 1] + 2 * (int)B[by + 1][bx] + (int)B[by + 1][bx + 1]) - ((int)B[by - 1][bx - 1] + 2 * (int)B[by - 1][bx] + (int)B[by - 1][bx + 1]);
            int my = ty + 1, mx = tx + 1;
            int m = MAG[my][mx];
            if (m > low) {
                int ax = abs(xs), ay = abs(ys) << 15;
                int tg22x = ax * 13573;
                bool keep;
                if (ay < tg22x) keep = (m > MAG[my][mx - 1] && m >= MAG[my][mx + 1]);
                else {
                    int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) keep = (m > MAG[my - 1][mx] && m >= MAG[my + 1][mx]);
                    else { int s = ((xs ^ ys) < 0) ? -1 : 1; keep = (m > MAG[my - 1][mx - s] && m > MAG[my + 1][mx + s]); }
                }
                if (keep) res = (m > high) ? 2 : 0;
            }
            if (map) dst[(size_t)y * W + x] = res;
        }
        if (cand_bits) {          // candidate (weak or strong) and strong planes, one word per wave
            const unsigned long long cm = __ballot(in && res != 1), sm = __ballot(in && res == 2);
            if ((threadIdx.x & 63) == 0 && y < H) {
                const size_t w = (size_t)H * Ww * blockIdx.z + (size_t)y * Ww + (x0 >> 6);
                cand_bits[w] = cm; strong_bits[w] = sm;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same for BINARY masks and the default 3x3 Gaussian, straight from the bit planes the morphology leaves (1 bit per pixel instead of a
// byte plane written and read back: 2 x K x H x W bytes of HBM traffic less, and no per-byte border arithmetic).  With mask values {0, 255}
// the blurred pixel is (255 c + 8) >> 4, c = the [1 2 1] x [1 2 1] weighted count of set pixels, i.e. a function of three 3-bit fields.
// Tile 64 x 32 outputs; LDS: 38 rows x 3 words of mask bits around the tile (BORDER_REFLECT_101 already applied to them), the blurred
// tile (halo 2, positions outside the image take the clamped position's value = Sobel's BORDER_REPLICATE), the magnitudes (halo 1).
// ------------------------------------------------------------------------------------------------
#define NB_TX 64
#define NB_TY 32
__global__ __launch_bounds__(256) void k_nms_bits3(const unsigned long long* __restrict__ mbits, int H, int W, int Ww, int low, int high,
                                                   unsigned long long* __restrict__ cand_bits, unsigned long long* __restrict__ strong_bits) {
    __shared__ unsigned long long R[NB_TY + 6][4];        // words x0/64 - 1 .. + 1 of rows y0 - 3 .. y0 + NB_TY + 2 (4th word: padding)
    __shared__ u8 B[NB_TY + 4][NB_TX + 4];
    __shared__ short MAG[NB_TY + 2][NB_TX + 2];
    const int bx = blockIdx.x, x0 = bx * NB_TX, y0 = blockIdx.y * NB_TY;
    const unsigned long long* src = mbits + (size_t)H * Ww * blockIdx.z;
    const int tid = threadIdx.x;
    // A tile whose whole neighbourhood (columns x0 - 3 .. x0 + 66 of the 38 rows) is uniformly 0 or uniformly 1 has a constant blurred
    // value, no gradient and no candidate: most tiles of a mask made of large regions leave here after one pass over 114 words.
    int mixed = 0;
    if (tid < (NB_TY + 6) * 3) {
        const int row = tid / 3, wj = tid % 3;
        const int y = reflect101(y0 - 3 + row, H), xw = bx - 1 + wj;
        const unsigned long long w = (xw >= 0 && xw < Ww) ? src[(size_t)y * Ww + xw] : 0ULL;
        R[row][wj] = w;
        // bits of this word that lie inside the image AND inside the neighbourhood
        unsigned long long need = wj == 0 ? (7ULL << 61) : (wj == 1 ? ~0ULL : 7ULL);
        if (xw < 0 || xw >= Ww) need = 0;
        else if (xw == Ww - 1 && (W & 63)) need &= (1ULL << (W & 63)) - 1ULL;
        const unsigned long long first = src[(size_t)reflect101(y0, H) * Ww + bx] & 1ULL ? ~0ULL : 0ULL;     // reference value: the tile's first pixel
        mixed = ((w ^ first) & need) != 0;
    }
    if (!__syncthreads_or(mixed)) {
        for (int ty = tid; ty < NB_TY; ty += 256) if (y0 + ty < H) { const size_t wd = (size_t)H * Ww * blockIdx.z + (size_t)(y0 + ty) * Ww + bx; cand_bits[wd] = 0ULL; strong_bits[wd] = 0ULL; }
        return;
    }
    if (tid < NB_TY + 6) {                                // reflected columns: x = -1, -2, -3 and x = W, W + 1, W + 2 where the window holds them
        auto getb = [&](int x) -> unsigned long long { const int p = x - (x0 - 64); return (R[tid][p >> 6] >> (p & 63)) & 1ULL; };
        auto setb = [&](int x, unsigned long long b) { const int p = x - (x0 - 64); if (p >= 0 && p < 192) R[tid][p >> 6] = (R[tid][p >> 6] & ~(1ULL << (p & 63))) | (b << (p & 63)); };
        if (bx == 0) for (int j = 1; j <= 3; j++) setb(-j, getb(j));
        if (x0 + NB_TX + 3 > W) for (int j = 0; j < 3; j++) setb(W + j, getb(W - 2 - j));
    }
    __syncthreads();
    // blurred tile with halo 2
    for (int i = tid; i < (NB_TY + 4) * (NB_TX + 4); i += 256) {
        const int ty = i / (NB_TX + 4), tx = i % (NB_TX + 4);
        const int y = y0 + ty - 2, x = x0 + tx - 2;
        const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
        const int r = yc - y0 + 3, p = xc - 1 - (x0 - 64);           // row of the centre in R; window bit of column xc - 1
        const int wi = p >> 6, sh = p & 63;
        auto field = [&](int rr) -> unsigned {
            unsigned long long v = R[rr][wi] >> sh;
            if (sh > 61) v |= R[rr][wi + 1] << (64 - sh);
            return (unsigned)v & 7u;
        };
        const unsigned LUT = 0u | (1u << 3) | (2u << 6) | (3u << 9) | (1u << 12) | (2u << 15) | (3u << 18) | (4u << 21);   // [1 2 1] . bits
        const unsigned c9 = ((LUT >> (3 * field(r - 1))) & 7u) + 2u * ((LUT >> (3 * field(r))) & 7u) + ((LUT >> (3 * field(r + 1))) & 7u);
        B[ty][tx] = (u8)((255u * c9 + 8u) >> 4);
    }
    __syncthreads();
    for (int i = tid; i < (NB_TY + 2) * (NB_TX + 2); i += 256) {     // magnitude with halo 1 (0 outside the image)
        const int ty = i / (NB_TX + 2), tx = i % (NB_TX + 2);
        const int y = y0 + ty - 1, x = x0 + tx - 1;
        int m = 0;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const int by = ty + 1, bx2 = tx + 1;
            const int gx = ((int)B[by - 1][bx2 + 1] + 2 * (int)B[by][bx2 + 1] + (int)B[by + 1][bx2 + 1]) - ((int)B[by - 1][bx2 - 1] + 2 * (int)B[by][bx2 - 1] + (int)B[by + 1][bx2 - 1]);
            const int gy = ((int)B[by + 1][bx2 - 1] + 2 * (int)B[by + 1][bx2] + (int)B[by + 1][bx2 + 1]) - ((int)B[by - 1][bx2 - 1] + 2 * (int)B[by - 1][bx2] + (int)B[by - 1][bx2 + 1]);
            m = abs(gx) + abs(gy);
        }
        MAG[ty][tx] = (short)m;
    }
    __syncthreads();
    for (int i = tid; i < NB_TY * NB_TX; i += 256) {                  // a wave = one row of the tile
        const int ty = i / NB_TX, tx = i % NB_TX;
        const int y = y0 + ty, x = x0 + tx;
        const bool in = y < H && x < W;
        int res = 1;
        if (in) {
            const int by = ty + 2, bx2 = tx + 2;
            const int xs = ((int)B[by - 1][bx2 + 1] + 2 * (int)B[by][bx2 + 1] + (int)B[by + 1][bx2 + 1]) - ((int)B[by - 1][bx2 - 1] + 2 * (int)B[by][bx2 - 1] + (int)B[by + 1][bx2 - 1]);
            const int ys = ((int)B[by + 1][bx2 - 1] + 2 * (int)B[by + 1][bx2] + (int)B[by + 1][bx2 + 1]) - ((int)B[by - 1][bx2 - 1] + 2 * (int)B[by - 1][bx2] + (int)B[by - 1][bx2 + 1]);
            const int my = ty + 1, mx = tx + 1;
            const int m = MAG[my][mx];
            if (m > low) {
                const int ax = abs(xs), ay = abs(ys) << 15;
                const int tg22x = ax * 13573;
                bool keep;
                if (ay < tg22x) keep = (m > MAG[my][mx - 1] && m >= MAG[my][mx + 1]);
                else {
                    const int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) keep = (m > MAG[my - 1][mx] && m >= MAG[my + 1][mx]);
                    else { const int sg = ((xs ^ ys) < 0) ? -1 : 1; keep = (m > MAG[my - 1][mx - sg] && m > MAG[my + 1][mx + sg]); }
                }
                if (keep) res = (m > high) ? 2 : 0;
            }
        }
        const unsigned long long cm = __ballot(in && res != 1), sm = __ballot(in && res == 2);
        if ((tid & 63) == 0 && y < H) {
            const size_t w = (size_t)H * Ww * blockIdx.z + (size_t)y * Ww + bx;
            cand_bits[w] = cm; strong_bits[w] = sm;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Generic 8-connected union-find CCL over "foreground" pixels.  Pixels are identified by their
// BLOCK-RASTER id  ((y>>1)*Wb + (x>>1))*4 + (y&1)*2 + (x&1): the root (minimum id) of a component
// then names the first 2x2 block, in block-raster order, that holds one of its pixels -- the label
// order block-based CCL scanners produce.  par[] is indexed by id, per layer.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int px_id(int y, int x, int Wb) { return (((y >> 1) * Wb + (x >> 1)) << 2) | ((y & 1) << 1) | (x & 1); }

__device__ __forceinline__ int uf_find(const int* L, int a) {
    int p = L[a];
    while (p != a) { a = p; p = L[a]; }
    return a;
}
__device__ __forceinline__ void uf_unite(int* L, int a, int b) {
    bool done;
    do {
        a = uf_find(L, a); b = uf_find(L, b);
        if (a < b) { int old = atomicMin(&L[b], a); done = (old == b); b = old; }
        else if (b < a) { int old = atomicMin(&L[a], b); done = (old == a); a = old; }
        else done = true;
    } while (!done);
}

// fgtest: 0 -> fg = (img != bg_value);  grid.z = layer
__global__ __launch_bounds__(256) void k_ccl_init(const u8* __restrict__ img, int* __restrict__ par, int H, int W, int bg_value) {
    const int Wb = (W + 1) >> 1, Hb = (H + 1) >> 1;
    const size_t plane = (size_t)H * W, pplane = (size_t)Wb * Hb * 4;
    const u8* s = img + plane * blockIdx.z; int* L = par + pplane * blockIdx.z;
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    // only foreground entries are ever read (union-find chains stay inside the foreground, every consumer tests the pixel first),
    // so the 4-byte parents of the background -- 98 % of the plane -- are not written at all
    if (s[(size_t)y * W + x] != bg_value) { const int id = px_id(y, x, Wb); L[id] = id; }
}
__global__ __launch_bounds__(256) void k_ccl_merge(const u8* __restrict__ img, int* __restrict__ par, int H, int W, int bg_value) {
    const int Wb = (W + 1) >> 1, Hb = (H + 1) >> 1;
    const size_t plane = (size_t)H * W, pplane = (size_t)Wb * Hb * 4;
    const u8* s = img + plane * blockIdx.z; int* L = par + pplane * blockIdx.z;
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    if (s[(size_t)y * W + x] == bg_value) return;
    int id = px_id(y, x, Wb);
    if (x > 0 && s[(size_t)y * W + x - 1] != bg_value) uf_unite(L, id, px_id(y, x - 1, Wb));
    if (y > 0) {
        const u8* up = s + (size_t)(y - 1) * W;
        if (x > 0 && up[x - 1] != bg_value) uf_unite(L, id, px_id(y - 1, x - 1, Wb));
        if (up[x] != bg_value) uf_unite(L, id, px_id(y - 1, x, Wb));
        if (x + 1 < W && up[x + 1] != bg_value) uf_unite(L, id, px_id(y - 1, x + 1, Wb));
    }
}
// plane-aware flatten
__global__ __launch_bounds__(256) void k_ccl_flatten2(const u8* __restrict__ img, int* __restrict__ par, int H, int W, int bg_value) {
    const int Wb = (W + 1) >> 1, Hb = (H + 1) >> 1;
    const size_t plane = (size_t)H * W, pplane = (size_t)Wb * Hb * 4;
    const u8* s = img + plane * blockIdx.z; int* L = par + pplane * blockIdx.z;
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    if (s[(size_t)y * W + x] == bg_value) return;
    const int id = px_id(y, x, Wb);
    L[id] = uf_find(L, id);
}
// ---- 16 pixels per thread (rows that are multiples of 16 wide): the planes are ~98 % background, and a thread whose sixteen bytes
// are all background returns after one 16-byte load; the rest is the same union-find as above.  mode 0: init, 1: merge, 2: flatten.
__global__ __launch_bounds__(256) void k_ccl16(const u8* __restrict__ img, int* __restrict__ par, int H, int W, int bg_value, int mode) {
    const int Wb = (W + 1) >> 1, Hb = (H + 1) >> 1, W16 = W >> 4;
    const size_t plane = (size_t)H * W, pplane = (size_t)Wb * Hb * 4;
    const u8* s = img + plane * blockIdx.z; int* L = par + pplane * blockIdx.z;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)H * W16) return;
    const int y = (int)(t / W16), x0 = (int)(t % W16) * 16;
    const uint4 cur = *reinterpret_cast<const uint4*>(s + (size_t)y * W + x0);
    const unsigned splat = 0x01010101u * (unsigned)bg_value;
    if (cur.x == splat && cur.y == splat && cur.z == splat && cur.w == splat) return;
    const unsigned cw[4] = {cur.x, cur.y, cur.z, cur.w};
    auto CUR = [&](int j) -> int { return (int)((cw[j >> 2] >> (8 * (j & 3))) & 0xffu); };
    if (mode == 0) { for (int j = 0; j < 16; j++) if (CUR(j) != bg_value) { const int id = px_id(y, x0 + j, Wb); L[id] = id; } return; }
    if (mode == 2) { for (int j = 0; j < 16; j++) if (CUR(j) != bg_value) { const int id = px_id(y, x0 + j, Wb); L[id] = uf_find(L, id); } return; }
    // merge: left neighbour and the three upper neighbours of every foreground pixel
    const int left = x0 > 0 ? (int)s[(size_t)y * W + x0 - 1] : bg_value;
    unsigned uw[4] = {splat, splat, splat, splat}; int ul = bg_value, ur = bg_value;
    if (y > 0) {
        const u8* up = s + (size_t)(y - 1) * W;
        const uint4 u = *reinterpret_cast<const uint4*>(up + x0);
        uw[0] = u.x; uw[1] = u.y; uw[2] = u.z; uw[3] = u.w;
        if (x0 > 0) ul = up[x0 - 1];
        if (x0 + 16 < W) ur = up[x0 + 16];
    }
    auto UP = [&](int j) -> int { return j < 0 ? ul : (j > 15 ? ur : (int)((uw[j >> 2] >> (8 * (j & 3))) & 0xffu)); };
    for (int j = 0; j < 16; j++) {
        if (CUR(j) == bg_value) continue;
        const int x = x0 + j, id = px_id(y, x, Wb);
        if ((j > 0 ? CUR(j - 1) : left) != bg_value) uf_unite(L, id, px_id(y, x - 1, Wb));
        if (y > 0) {
            if (UP(j - 1) != bg_value) uf_unite(L, id, px_id(y - 1, x - 1, Wb));
            if (UP(j) != bg_value) uf_unite(L, id, px_id(y - 1, x, Wb));
            if (UP(j + 1) != bg_value) uf_unite(L, id, px_id(y - 1, x + 1, Wb));
        }
    }
}
// components of bit planes (one bit per pixel, 64 per word, blockIdx.z = layer): a thread owns a word, returns at once when it is
// empty and walks its set bits otherwise.  Same ids (block raster) and the same union-find as above.  mode 0 init, 1 merge, 2 flatten.
__global__ __launch_bounds__(256) void k_ccl_bits(const unsigned long long* __restrict__ bits, int* __restrict__ par, int H, int W, int Ww, int mode) {
    const size_t nw = (size_t)H * Ww, wi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (wi >= nw) return;
    const unsigned long long* b = bits + nw * blockIdx.z;
    unsigned long long m = b[wi];
    if (!m) return;
    const int Wb = (W + 1) >> 1, Hb = (H + 1) >> 1;
    int* L = par + (size_t)Wb * Hb * 4 * blockIdx.z;
    const int y = (int)(wi / Ww), xw = (int)(wi % Ww), x0 = xw * 64;
    if (mode != 1) {
        while (m) { const int j = __ffsll((long long)m) - 1; m &= m - 1; const int id = px_id(y, x0 + j, Wb); L[id] = mode == 0 ? id : uf_find(L, id); }
        return;
    }
    const unsigned long long left = xw > 0 ? b[wi - 1] : 0ULL;
    unsigned long long U = 0, UL = 0, UR = 0;
    if (y > 0) { U = b[wi - Ww]; if (xw > 0) UL = b[wi - Ww - 1]; if (xw + 1 < Ww) UR = b[wi - Ww + 1]; }
    const unsigned long long hasW = (m << 1) | (left >> 63), hasNW = (U << 1) | (UL >> 63), hasNE = (U >> 1) | (UR << 63);
    while (m) {
        const int j = __ffsll((long long)m) - 1; m &= m - 1;
        const int x = x0 + j, id = px_id(y, x, Wb);
        if ((hasW >> j) & 1ULL) uf_unite(L, id, px_id(y, x - 1, Wb));
        if ((hasNW >> j) & 1ULL) uf_unite(L, id, px_id(y - 1, x - 1, Wb));
        if ((U >> j) & 1ULL) uf_unite(L, id, px_id(y - 1, x, Wb));
        if ((hasNE >> j) & 1ULL) uf_unite(L, id, px_id(y - 1, x + 1, Wb));
    }
}
int orip_ccl_bits(orip_ctx* c, const unsigned long long* bits, int* par, int K) {
    const int H = c->H, W = c->W, Ww = (W + 63) >> 6;
    dim3 g((unsigned)cdiv((int64_t)H * Ww, 256), 1, K), block(256);
    { ProfScope ps(c, "k_ccl_init"); hipLaunchKernelGGL(k_ccl_bits, g, block, 0, LN(c).stream, bits, par, H, W, Ww, 0); }
    { ProfScope ps(c, "k_ccl_bits"); hipLaunchKernelGGL(k_ccl_bits, g, block, 0, LN(c).stream, bits, par, H, W, Ww, 1); }
    { ProfScope ps(c, "k_ccl_flatten"); hipLaunchKernelGGL(k_ccl_bits, g, block, 0, LN(c).stream, bits, par, H, W, Ww, 2); }
    HIPC(c, hipGetLastError());
    return 0;
}

int orip_ccl(orip_ctx* c, const u8* img, int* par, int K, int bg_value) {
    int H = c->H, W = c->W;
    dim3 grid(cdiv(W, 64), cdiv(H, 4), K), block(256);
    if ((W & 15) == 0 && !getenv("ORIP_CCL_BYTES")) {
        dim3 g16((unsigned)cdiv((int64_t)H * (W >> 4), 256), 1, K);
        { ProfScope ps(c, "k_ccl_init"); hipLaunchKernelGGL(k_ccl16, g16, block, 0, LN(c).stream, img, par, H, W, bg_value, 0); }
        { ProfScope ps(c, "k_ccl_merge"); hipLaunchKernelGGL(k_ccl16, g16, block, 0, LN(c).stream, img, par, H, W, bg_value, 1); }
        { ProfScope ps(c, "k_ccl_flatten"); hipLaunchKernelGGL(k_ccl16, g16, block, 0, LN(c).stream, img, par, H, W, bg_value, 2); }
        HIPC(c, hipGetLastError());
        return 0;
    }
    { ProfScope ps(c, "k_ccl_init"); hipLaunchKernelGGL(k_ccl_init, grid, block, 0, LN(c).stream, img, par, H, W, bg_value); }
    { ProfScope ps(c, "k_ccl_merge"); hipLaunchKernelGGL(k_ccl_merge, grid, block, 0, LN(c).stream, img, par, H, W, bg_value); }
    { ProfScope ps(c, "k_ccl_flatten"); hipLaunchKernelGGL(k_ccl_flatten2, grid, block, 0, LN(c).stream, img, par, H, W, bg_value); }
    HIPC(c, hipGetLastError());
    return 0;
}

// hysteresis: mark roots that hold a strong pixel, then edge = candidate && marked(root)
__global__ __launch_bounds__(256) void k_hyst_mark(const u8* __restrict__ map, const int* __restrict__ par, u8* __restrict__ strong_root, int H, int W) {
    const int Wb = (W + 1) >> 1, Hb = (H + 1) >> 1;
    const size_t plane = (size_t)H * W, pplane = (size_t)Wb * Hb * 4;
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    if (map[plane * blockIdx.z + (size_t)y * W + x] != 2) return;
    int root = par[pplane * blockIdx.z + px_id(y, x, Wb)];
    strong_root[pplane * blockIdx.z + root] = 1;
}
__global__ __launch_bounds__(256) void k_hyst_out(const u8* __restrict__ map, const int* __restrict__ par, const u8* __restrict__ strong_root,
                                                   u8* __restrict__ edges, int H, int W) {
    const int Wb = (W + 1) >> 1, Hb = (H + 1) >> 1;
    const size_t plane = (size_t)H * W, pplane = (size_t)Wb * Hb * 4;
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    size_t o = plane * blockIdx.z + (size_t)y * W + x;
    u8 v = 0;
    if (map[o] != 1) { int root = par[pplane * blockIdx.z + px_id(y, x, Wb)]; v = strong_root[pplane * blockIdx.z + root] ? 255 : 0; }
    edges[o] = v;
}

// hysteresis on bit planes: roots of components that hold a strong pixel, then edge = candidate whose root is marked
__global__ __launch_bounds__(256) void k_hyst_bits_mark(const unsigned long long* __restrict__ strong_bits, const int* __restrict__ par, u8* __restrict__ strong_root, int H, int W, int Ww) {
    const size_t nw = (size_t)H * Ww, wi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (wi >= nw) return;
    unsigned long long m = strong_bits[nw * blockIdx.z + wi];
    if (!m) return;
    const int Wb = (W + 1) >> 1, Hb = (H + 1) >> 1; const size_t pplane = (size_t)Wb * Hb * 4;
    const int y = (int)(wi / Ww), x0 = (int)(wi % Ww) * 64;
    while (m) { const int j = __ffsll((long long)m) - 1; m &= m - 1; strong_root[pplane * blockIdx.z + par[pplane * blockIdx.z + px_id(y, x0 + j, Wb)]] = 1; }
}
__global__ __launch_bounds__(256) void k_hyst_bits_out(const unsigned long long* __restrict__ cand_bits, const int* __restrict__ par, const u8* __restrict__ strong_root,
                                                        unsigned long long* __restrict__ edge_bits, int H, int W, int Ww) {
    const size_t nw = (size_t)H * Ww, wi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (wi >= nw) return;
    unsigned long long m = cand_bits[nw * blockIdx.z + wi], e = 0;
    if (m) {
        const int Wb = (W + 1) >> 1, Hb = (H + 1) >> 1; const size_t pplane = (size_t)Wb * Hb * 4;
        const int y = (int)(wi / Ww), x0 = (int)(wi % Ww) * 64;
        while (m) { const int j = __ffsll((long long)m) - 1; m &= m - 1; if (strong_root[pplane * blockIdx.z + par[pplane * blockIdx.z + px_id(y, x0 + j, Wb)]]) e |= 1ULL << j; }
    }
    edge_bits[nw * blockIdx.z + wi] = e;
}
__global__ __launch_bounds__(256) void k_bits_to_bytes03(const unsigned long long* __restrict__ bits, u8* __restrict__ dst, int H, int W, int Ww) {
    const size_t nw = (size_t)H * Ww, w0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (w0 >= nw) return;
    const unsigned long long* b = bits + nw * blockIdx.z; u8* d = dst + (size_t)H * W * blockIdx.z;
    const int lane = threadIdx.x & 63;
    const unsigned long long mine = (w0 + lane < nw) ? b[w0 + lane] : 0ULL;
    for (int j = 0; j < 64; j++) {
        const size_t wi = w0 + j; if (wi >= nw) break;
        const unsigned long long v = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), j) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, j);
        const int y = (int)(wi / Ww), xx = (int)(wi % Ww) * 64 + lane;
        if (xx < W) d[(size_t)y * W + xx] = ((v >> lane) & 1ULL) ? 255 : 0;
    }
}

extern "C" int orip_detect_edges(orip_ctx* c, int morph_k, int open_iters, int close_iters, int gauss_k, int low, int high) {
    orip_enter(c);
    if (!c->masks.p || c->K < 1) ORIP_FAIL(c, "no masks resident (run orip_extract_layers or orip_set_masks)");
    if (gauss_k != 3 && gauss_k != 5 && gauss_k != 7) ORIP_FAIL(c, "GaussianBlur kernel size %d unsupported (3, 5, 7)", gauss_k);
    int H = c->H, W = c->W, K = c->K; size_t plane = (size_t)H * W;
    if (low > high) std::swap(low, high);
    HIPC(c, c->edges.ensure(plane * K));
    HIPC(c, c->tmpB.ensure(plane * K));   // morphed masks
    HIPC(c, c->tmpC.ensure(plane * K));   // NMS map
    // binary masks + 3x3 Gaussian: the NMS kernel reads the morphed bit planes, no byte plane in between
    const bool nms_from_bits = gauss_k == 3 && W >= 8 && H >= 8 && !getenv("ORIP_CCL_BYTES") && !getenv("ORIP_NMS_BYTES");
    c->morphed_bits = nullptr;
    ORIP_TRY(orip_morph_open_close(c, c->masks.as<u8>(), c->tmpB.as<u8>(), K, 2, morph_k, open_iters, close_iters, false, !nms_from_bits));
    dim3 grid(cdiv(W, ET_X), cdiv(H, ET_Y), K), block(256);
    int Wb = (W + 1) >> 1, Hb = (H + 1) >> 1; size_t pplane = (size_t)Wb * Hb * 4;
    HIPC(c, c->tmpD.ensure(pplane * K * sizeof(int)));
    HIPC(c, LN(c).tmpE.ensure(pplane * K));
    c->edge_bits = nullptr;
    if (!getenv("ORIP_CCL_BYTES")) {
        // candidates are ~2 % of the pixels: NMS leaves two bit planes (candidate, strong); components, strong roots and edges are
        // computed from words; the edge bit planes stay for stage 04's thinning
        const int Ww = (W + 63) >> 6; const size_t nw = (size_t)H * Ww;
        HIPC(c, LN(c).vtmp[11].ensure(nw * K * 16 + 64));
        HIPC(c, LN(c).vtmp[10].ensure(nw * K * 16 + 64));
        unsigned long long* cand = LN(c).vtmp[11].as<unsigned long long>(); unsigned long long* strong = cand + nw * K;
        unsigned long long* ebits = LN(c).vtmp[10].as<unsigned long long>();
        if (nms_from_bits && c->morphed_bits) {
            ProfScope ps(c, "k_blur_sobel_nms");
            hipLaunchKernelGGL(k_nms_bits3, dim3(cdiv(W, NB_TX), cdiv(H, NB_TY), K), block, 0, LN(c).stream, (const unsigned long long*)c->morphed_bits, H, W, Ww, low, high, cand, strong);
        } else { ProfScope ps(c, "k_blur_sobel_nms"); hipLaunchKernelGGL(k_blur_sobel_nms, grid, block, 0, LN(c).stream, c->tmpB.as<u8>(), (u8*)nullptr, H, W, gauss_k, low, high, cand, strong, Ww); }
        c->morphed_bits = nullptr;
        ORIP_TRY(orip_ccl_bits(c, cand, c->tmpD.as<int>(), K));
        HIPC(c, hipMemsetAsync(LN(c).tmpE.p, 0, pplane * K, LN(c).stream));
        dim3 gw((unsigned)cdiv((int64_t)nw, 256), 1, K);
        { ProfScope ps(c, "k_hyst_mark"); hipLaunchKernelGGL(k_hyst_bits_mark, gw, block, 0, LN(c).stream, strong, c->tmpD.as<int>(), LN(c).tmpE.as<u8>(), H, W, Ww); }
        { ProfScope ps(c, "k_hyst_out"); hipLaunchKernelGGL(k_hyst_bits_out, gw, block, 0, LN(c).stream, cand, c->tmpD.as<int>(), LN(c).tmpE.as<u8>(), ebits, H, W, Ww); }
        if ((W & 63) == 0 && !getenv("ORIP_PACK_BYTES")) hipLaunchKernelGGL(k_bits_expand16, dim3((unsigned)cdiv((int64_t)H * Ww * 4, 256), 1, K), block, 0, LN(c).stream, ebits, c->edges.as<u8>(), (size_t)H * Ww);
        else hipLaunchKernelGGL(k_bits_to_bytes03, gw, block, 0, LN(c).stream, ebits, c->edges.as<u8>(), H, W, Ww);
        HIPC(c, hipGetLastError());
        c->edge_bits = ebits;
        return 0;
    }
    { ProfScope ps(c, "k_blur_sobel_nms"); hipLaunchKernelGGL(k_blur_sobel_nms, grid, block, 0, LN(c).stream, c->tmpB.as<u8>(), c->tmpC.as<u8>(), H, W, gauss_k, low, high, (unsigned long long*)nullptr, (unsigned long long*)nullptr, 0); }
    HIPC(c, hipGetLastError());
    ORIP_TRY(orip_ccl(c, c->tmpC.as<u8>(), c->tmpD.as<int>(), K, 1));
    HIPC(c, hipMemsetAsync(LN(c).tmpE.p, 0, pplane * K, LN(c).stream));
    dim3 g2(cdiv(W, 64), cdiv(H, 4), K);
    { ProfScope ps(c, "k_hyst_mark"); hipLaunchKernelGGL(k_hyst_mark, g2, block, 0, LN(c).stream, c->tmpC.as<u8>(), c->tmpD.as<int>(), LN(c).tmpE.as<u8>(), H, W); }
    { ProfScope ps(c, "k_hyst_out"); hipLaunchKernelGGL(k_hyst_out, g2, block, 0, LN(c).stream, c->tmpC.as<u8>(), c->tmpD.as<int>(), LN(c).tmpE.as<u8>(), c->edges.as<u8>(), H, W); }
    HIPC(c, hipGetLastError());
    return 0;
}

extern "C" int orip_get_edges(orip_ctx* c, int layer, uint8_t* out) {
    orip_enter(c);
    if (!c->edges.p || layer < 0 || layer >= c->K) ORIP_FAIL(c, "no edges for layer %d", layer);
    size_t plane = (size_t)c->H * c->W;
    HIPC(c, hipMemcpyAsync(out, c->edges.as<u8>() + plane * layer, plane, hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    return 0;
}
extern "C" int orip_set_edges(orip_ctx* c, const uint8_t* edges, int K, int H, int W) {
    orip_enter(c);
    c->edge_bits = nullptr;
    if (K < 1 || K > ORIP_MAX_LAYERS || H <= 0 || W <= 0) ORIP_FAIL(c, "bad shape");
    c->H = H; c->W = W; c->K = K;
    HIPC(c, c->edges.ensure((size_t)H * W * K));
    HIPC(c, hipMemcpyAsync(c->edges.p, edges, (size_t)H * W * K, hipMemcpyHostToDevice, LN(c).stream));
    return 0;
}
