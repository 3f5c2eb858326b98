// csrc/raster04.hip -- stage 04 (04_find_contours.py vectorize_layer, 04:214-230) on gfx950:
//   k_thin_sub        : one Zhang-Suen sub-iteration with the reference's rotated neighbour numbering (04:50-93)
//   CCL               : union-find kernels of raster03.hip (component order = block-raster)
//   k_skel_state      : per-pixel state byte (fg, endpoint deg==1, junction deg>=3)          (04:128-130)
//   k_compact_*       : wavefront ballot / prefix-sum compaction of skeleton pixels in raster order (04:144,174)
//   radix sort        : rocPRIM stable sort of (layer, component root) keys -> per-component pixel lists
//   k_trace / k_write_walks : the centerline walker (walker.h), exact serial semantics (04:137-205): one wave per component
//                       runs twice (count, then write); the guard-bounded "bounce" tails of phase-2 walks are
//                       detected as cycles of the (prev,cur) state and written by k_expand_cycles in parallel.
#include "orip_ctx.h"
#include "walker.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cstdlib>

int orip_ccl(orip_ctx* c, const u8* img, int* par, int K, int bg_value);
int orip_ccl_bits(orip_ctx* c, const unsigned long long* bits, int* par, int K);

// ------------------------------------------------------------------------------------------------
// Thinning.  P2..P9 offsets (dy,dx) derived from the _shift() arguments at 04:53-55.
// ------------------------------------------------------------------------------------------------
#ifdef ORIP_VARIANTS      // replaced variant (ORIP_THIN_BYTES): variants build only (make variants)
__global__ __launch_bounds__(256) void k_thin_sub(const u8* __restrict__ src, u8* __restrict__ dst, int H, int W, int sub, int* __restrict__ changed) {
    const size_t plane = (size_t)H * W;
    const u8* s = src + plane * blockIdx.z; u8* d = dst + plane * blockIdx.z;
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    size_t o = (size_t)y * W + x;
    u8 v = s[o];
    if (v) {
        auto g = [&](int dy, int dx) -> int { int yy = y + dy, xx = x + dx; return (yy >= 0 && yy < H && xx >= 0 && xx < W && s[(size_t)yy * W + xx]) ? 1 : 0; };
        int P2 = g(1, 0), P3 = g(1, -1), P4 = g(0, -1), P5 = g(-1, -1), P6 = g(-1, 0), P7 = g(-1, 1), P8 = g(0, 1), P9 = g(1, 1);
        int Bn = P2 + P3 + P4 + P5 + P6 + P7 + P8 + P9;
        int A = (!P2 && P3) + (!P3 && P4) + (!P4 && P5) + (!P5 && P6) + (!P6 && P7) + (!P7 && P8) + (!P8 && P9) + (!P9 && P2);
        bool cnd = sub == 0 ? (P2 * P4 * P6 == 0 && P4 * P6 * P8 == 0) : (P2 * P4 * P8 == 0 && P2 * P6 * P8 == 0);
        if (A == 1 && Bn >= 2 && Bn <= 6 && cnd) { v = 0; *changed = 1; }
    }
    d[o] = v ? 255 : 0;
}
#endif

// ------------------------------------------------------------------------------------------------
// state byte (walker.h): ST_FG, ST_VIS, ST_END (deg == 1), ST_JUN (deg >= 3)
// ------------------------------------------------------------------------------------------------
#ifdef ORIP_VARIANTS      // replaced variant (ORIP_THIN_BYTES): variants build only (make variants)
__global__ __launch_bounds__(256) void k_skel_state(const u8* __restrict__ skel, u8* __restrict__ st, int H, int W) {
    const size_t plane = (size_t)H * W;
    const u8* s = skel + plane * blockIdx.z; u8* d = st + plane * blockIdx.z;
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    size_t o = (size_t)y * W + x;
    u8 v = 0;
    if (s[o]) {
        int deg = 0;
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                if (!dy && !dx) continue;
                int yy = y + dy, xx = x + dx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W && s[(size_t)yy * W + xx]) deg++;
            }
        v = ST_FG | (deg == 1 ? ST_END : 0) | (deg >= 3 ? ST_JUN : 0) | (deg == 2 ? ST_DEG2 : 0);
    }
    d[o] = v;
}
#endif

// ------------------------------------------------------------------------------------------------
// Ordered compaction (raster order inside a layer, layers in order): ballot + popcount inside the wave,
// LDS across the 4 waves of a block, block offsets from an exclusive scan of per-block counts.
// One block = 1024 consecutive pixels of the flattened [K,H,W] array (4 per thread via one dword load).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact_count(const u8* __restrict__ st, int64_t n, unsigned* __restrict__ counts) {
    __shared__ unsigned wsum[4];
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    unsigned cnt = 0;
    if (i + 3 < n) { uint32_t w = *reinterpret_cast<const uint32_t*>(st + i); cnt = __popc(w & 0x80808080u); }        // ST_FG of four state bytes
    else for (int j = 0; j < 4; j++) if (i + j < n && (st[i + j] & ST_FG)) cnt++;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(256) void k_compact_write(const u8* __restrict__ st, const int* __restrict__ par, int64_t n, int H, int W,
                                                       const unsigned* __restrict__ block_off, unsigned* __restrict__ keys, unsigned* __restrict__ lin) {
    __shared__ unsigned wsum[4];
    const int64_t plane = (int64_t)H * W;
    const int Wb = (W + 1) >> 1, Hb = (H + 1) >> 1; const int64_t pplane = (int64_t)Wb * Hb * 4;
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    unsigned f[4]; unsigned cnt = 0;
    for (int j = 0; j < 4; j++) { f[j] = (i + j < n && (st[i + j] & ST_FG)) ? 1u : 0u; cnt += f[j]; }
    // inclusive scan of cnt across the wave via shuffles, then across waves via LDS
    unsigned inc = cnt;
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned base = block_off[blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) base += wsum[w];
    unsigned pos = base + inc - cnt;
    for (int j = 0; j < 4; j++) {
        if (!f[j]) continue;
        int64_t g = i + j; int layer = (int)(g / plane); int64_t p = g - (int64_t)layer * plane;
        int y = (int)(p / W), x = (int)(p % W);
        int id = (((y >> 1) * Wb + (x >> 1)) << 2) | ((y & 1) << 1) | (x & 1);
        unsigned root = (unsigned)par[pplane * layer + id];
        keys[pos] = ((unsigned)layer << 26) | root;
        lin[pos] = (unsigned)p;
        pos++;
    }
}

// The same compaction from the thinned BIT planes ([K][H][Ww] words; a set bit = a skeleton pixel = ST_FG in the state plane): a thread owns a word, a
// block 256 consecutive words of the flattened planes, i.e. the same (layer, row, column) order.  The byte form read 134 MB of state bytes at 4096^2 x 8,
// four per thread (0.43 ms in front of the walks); the planes are 16 MB and almost empty.
__global__ __launch_bounds__(256) void k_compact_count_bits(const unsigned long long* __restrict__ bits, size_t nwords, unsigned* __restrict__ counts) {
    __shared__ unsigned wsum[4];
    const size_t wi = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned cnt = wi < nwords ? (unsigned)__popcll(bits[wi]) : 0u;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(256) void k_compact_write_bits(const unsigned long long* __restrict__ bits, const int* __restrict__ par, size_t nwords, size_t nw, int H, int W, int Ww,
                                                            const unsigned* __restrict__ block_off, unsigned* __restrict__ keys, unsigned* __restrict__ lin) {
    __shared__ unsigned wsum[4];
    const int Wb = (W + 1) >> 1, Hb = (H + 1) >> 1; const int64_t pplane = (int64_t)Wb * Hb * 4;
    const size_t wi = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long m = wi < nwords ? bits[wi] : 0ULL;
    const unsigned cnt = (unsigned)__popcll(m);
    unsigned inc = cnt;
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (!m) return;
    unsigned pos = block_off[blockIdx.x] + inc - cnt;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) pos += wsum[w];
    const int layer = (int)(wi / nw); const size_t wl = wi - (size_t)layer * nw;
    const int y = (int)(wl / Ww), x0 = (int)(wl % Ww) * 64;
    while (m) {
        const int j = __ffsll((long long)m) - 1; m &= m - 1;
        const int x = x0 + j;
        const int id = (((y >> 1) * Wb + (x >> 1)) << 2) | ((y & 1) << 1) | (x & 1);
        keys[pos] = ((unsigned)layer << 26) | (unsigned)par[pplane * layer + id];
        lin[pos] = (unsigned)((int64_t)y * W + x);
        pos++;
    }
}

__global__ __launch_bounds__(256) void k_heads(const unsigned* __restrict__ keys, int64_t m, unsigned* __restrict__ head) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// comp_start[c] = first element of component c; comp_start[nc] = m
__global__ __launch_bounds__(256) void k_comp_starts(const unsigned* __restrict__ head, const unsigned* __restrict__ head_scan, int64_t m,
                                                     unsigned* __restrict__ comp_start, unsigned nc) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) comp_start[nc] = (unsigned)m;
    if (i >= m) return;
    if (head[i]) comp_start[head_scan[i]] = (unsigned)i;
}
__global__ __launch_bounds__(256) void k_gather_head_layers(const unsigned* __restrict__ keys, const unsigned* __restrict__ cs, unsigned nc, unsigned* __restrict__ out) {
    unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < nc) out[i] = keys[cs[i]] >> 26;
}

// ------------------------------------------------------------------------------------------------
// Walker.  NEIGH8 order (dx,dy) from 04:12.
// ------------------------------------------------------------------------------------------------
#include "walker.h"

// one wavefront per component; wave i takes component comp_order[i] (largest components first, so the long serial
// chains start immediately and the small ones fill in behind them)
__global__ __launch_bounds__(64) void k_trace(WalkArgs A) {
    unsigned i = blockIdx.x;
    if (i >= A.nc) return;
    trace_component(A, A.comp_order ? A.comp_order[i] : i);
}
// one wavefront per chunk of `chunk` output points
__global__ __launch_bounds__(64) void k_write_walks(WalkArgs A, int layer, const unsigned* __restrict__ kept_slots, const unsigned long long* __restrict__ kept_off, unsigned n_kept,
                                                    unsigned long long total, unsigned chunk) {
    for (unsigned long long ci = blockIdx.x; ci * chunk < total; ci += gridDim.x) {
        const unsigned long long p0 = ci * chunk, p1 = p0 + chunk < total ? p0 + chunk : total;
        write_chunk(A, layer, kept_slots, kept_off, n_kept, p0, p1);
    }
}
// lengths of the layer's walks (slots [slot0, slot0 + n)); walks that ended inside a recorded trajectory get their closing point settled here.
// opc: own points (n_own + 1) << 32 | tail pieces of a kept walk (one scan sizes both arrays of the walk-coded form)
__global__ void k_totals_flag(const int* __restrict__ over, unsigned* __restrict__ pad_of_total) { *pad_of_total = (unsigned)*over; }
struct WSum {       // what one walk slot adds to the layer's list: one scan sizes and places everything
    unsigned long long pts; unsigned paths, own, pieces, pad;
    __host__ __device__ WSum operator+(const WSum& o) const { WSum r; r.pts = pts + o.pts; r.paths = paths + o.paths; r.own = own + o.own; r.pieces = pieces + o.pieces; r.pad = 0; return r; }
};
__global__ __launch_bounds__(256) void k_winfo_lens(WalkArgs A, unsigned slot0, unsigned n, WSum* __restrict__ ws) {
    unsigned i = blockIdx.x * 256 + threadIdx.x;
    WSum s; s.pts = 0; s.paths = 0; s.own = 0; s.pieces = 0; s.pad = 0;
    if (i < n && !*A.overflow) {                       // (logs of a trace that ran out of room are not followed: the host retries it)
        WalkInfo w = A.winfo[slot0 + i];
        if (w.flags & 2u) { walk_close_tail(A, PlainReader(), slot0 + i, w); A.winfo[slot0 + i] = w; }
        if (w.len_kept) { s.pts = w.len_kept; s.paths = 1u; s.own = w.n_own + 1u; s.pieces = vwalk_pieces(A.logbuf, PlainReader(), w, [](unsigned, unsigned, unsigned, unsigned) {}); }
    }
    if (i <= n) ws[i] = s;
}
// pixel of every log entry in use, and the list of those entries (any order: blocks take their ranges with one atomic each).
// blockIdx.x = component of the layer, blockIdx.y = slice of its entries
__global__ __launch_bounds__(256) void k_ent_fill(WalkArgs A, unsigned c0, unsigned log_shift, int2* __restrict__ lxy, unsigned* __restrict__ ent_idx, unsigned* __restrict__ cnt, unsigned cap) {
    __shared__ unsigned base_s;
    if (*A.overflow) return;
    const unsigned c = c0 + blockIdx.x;
    const unsigned used = A.log_used[c];
    const unsigned per = (used + gridDim.y - 1) / gridDim.y, t0 = blockIdx.y * per, t1 = min(used, t0 + per);
    if (t0 >= t1) return;
    if (threadIdx.x == 0) base_s = atomicAdd(cnt, t1 - t0);
    __syncthreads();
    const unsigned base = base_s;
    const unsigned lb = A.cap_factor * A.comp_start[c] + 64u * c;           // first entry of the component (walker.h: trace_component)
    const unsigned W = (unsigned)A.W;
    for (unsigned t = t0 + threadIdx.x; t < t1; t += 256) {
        const unsigned e = lb + t, lin = A.logbuf[4ull * e] >> 3;
        const unsigned y = lin / W;
        lxy[e - log_shift] = make_int2((int)(lin - y * W), (int)y);
        if (base + (t - t0) < cap) ent_idx[base + (t - t0)] = e - log_shift;
    }
}
// the walk-coded form of the layer's contours: one VWalk per kept walk (path order = slot order), its tail pieces, its offset in the list
__global__ __launch_bounds__(256) void k_vwalk_fill(WalkArgs A, unsigned slot0, unsigned n, const WSum* __restrict__ wo /* exclusive scan */, unsigned log_shift,
                                                     VWalk* __restrict__ vw, VPiece* __restrict__ vp, unsigned* __restrict__ kept_slots, int64_t* __restrict__ off) {
    unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i == n) off[wo[n].paths] = (int64_t)wo[n].pts;
    if (i >= n) return;
    const WSum o = wo[i];
    if (wo[i + 1].paths == o.paths) return;            // not kept
    const WalkInfo w = A.winfo[slot0 + i];
    VWalk v; v.own_off = o.own; v.n_own = w.n_own; v.piece_off = o.pieces; v.len = w.len_kept; v.flags = w.flags & 1u; v.pad0 = v.pad1 = 0;
    VPiece* mine = vp + v.piece_off;
    v.n_piece = vwalk_pieces(A.logbuf, PlainReader(), w, [&](unsigned j, unsigned u0, unsigned ent, unsigned lam) { VPiece q; q.u0 = u0; q.ent = ent - log_shift; q.lam = lam; q.magic = lam > 1u ? (unsigned)(0x100000000ull / lam) : 0xffffffffu; mine[j] = q; });
    const unsigned pi = o.paths;
    vw[pi] = v; kept_slots[pi] = slot0 + i; off[pi] = (int64_t)o.pts;
}
// own points of the kept walks, one wavefront per chunk
__global__ __launch_bounds__(64) void k_vown(WalkArgs A, const unsigned* __restrict__ kept_slots, const VWalk* __restrict__ vw, unsigned n_kept, int2* __restrict__ own, unsigned total, unsigned chunk) {
    for (unsigned long long ci = blockIdx.x; ci * chunk < total; ci += gridDim.x) {
        const unsigned q0 = (unsigned)(ci * chunk), q1 = q0 + chunk < total ? q0 + chunk : total;
        own_chunk(A, kept_slots, vw, n_kept, own, q0, q1);
    }
}

// ------------------------------------------------------------------------------------------------
// rocPRIM helpers
// ------------------------------------------------------------------------------------------------
template <class T>
static int excl_scan(orip_ctx* c, const T* in, T* out, size_t n, DBuf& tmp) {
    size_t bytes = 0;
    HIPC(c, rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), LN(c).stream));
    HIPC(c, tmp.ensure(bytes + 16));
    HIPC(c, rocprim::exclusive_scan(tmp.p, bytes, in, out, T(0), n, rocprim::plus<T>(), LN(c).stream));
    return 0;
}

// ---- thinning_zhangsuen (04:35-99) and the state bytes on bit planes: one bit per pixel, 64 pixels per word, blockIdx.z = layer.
// The reference numbers the neighbours from the south (P2 = (y+1, x), then clockwise as seen with y down: SW, W, NW, N, NE, E, SE);
// A(p) counts transitions around the same cycle wherever it starts, B(p) is symmetric, only the two product conditions differ from
// the usual orientation.  Bit-sliced evaluation as in stage 08-B (vector08.hip: k_zs_bits).
__global__ __launch_bounds__(256) void k_bytes_to_bits04(const u8* __restrict__ src, unsigned long long* __restrict__ bits, int H, int W, int Ww) {
    const size_t nw = (size_t)H * Ww, w0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (w0 >= nw) return;
    const u8* s = src + (size_t)H * W * blockIdx.z; unsigned long long* b = bits + nw * blockIdx.z;
    const int lane = threadIdx.x & 63;
    unsigned long long mine = 0;
    for (int j = 0; j < 64; j++) {
        const size_t wi = w0 + j; bool fg = false;
        if (wi < nw) { const int y = (int)(wi / Ww), x = (int)(wi % Ww) * 64 + lane; fg = x < W && s[(size_t)y * W + x] != 0; }
        const unsigned long long m = __ballot(fg);
        if (lane == j) mine = m;
    }
    if (w0 + lane < nw) b[w0 + lane] = mine;
}
struct Nb8 { unsigned long long N, NE, E, SE, S, SW, W, NW; };
__device__ __forceinline__ Nb8 neighbour_planes(const unsigned long long* __restrict__ s, int H, int Ww, int y, int xw, unsigned long long M) {
    auto W64 = [&](int yy, int xx) -> unsigned long long { return (yy < 0 || yy >= H || xx < 0 || xx >= Ww) ? 0ULL : s[(size_t)yy * Ww + xx]; };
    const unsigned long long U = W64(y - 1, xw), UL = W64(y - 1, xw - 1), UR = W64(y - 1, xw + 1), ML = W64(y, xw - 1), MR = W64(y, xw + 1);
    const unsigned long long D = W64(y + 1, xw), DL = W64(y + 1, xw - 1), DR = W64(y + 1, xw + 1);
    Nb8 n; n.N = U; n.NE = (U >> 1) | (UR << 63); n.E = (M >> 1) | (MR << 63); n.SE = (D >> 1) | (DR << 63);
    n.S = D; n.SW = (D << 1) | (DL >> 63); n.W = (M << 1) | (ML >> 63); n.NW = (U << 1) | (UL >> 63);
    return n;
}
// bit-sliced neighbour count: b0 ones, b1 twos, b2 fours, b3 eights
__device__ __forceinline__ void count8(const Nb8& n, unsigned long long& b0, unsigned long long& b1, unsigned long long& b2, unsigned long long& b3) {
    auto FA = [](unsigned long long a, unsigned long long b, unsigned long long c, unsigned long long& sum, unsigned long long& carry) { const unsigned long long t = a ^ b; sum = t ^ c; carry = (a & b) | (t & c); };
    unsigned long long s1, c1, s2, c2, s4, c4, s5, c5;
    FA(n.N, n.NE, n.E, s1, c1); FA(n.SE, n.S, n.SW, s2, c2);
    const unsigned long long s3 = n.W ^ n.NW, c3 = n.W & n.NW;
    FA(s1, s2, s3, s4, c4); FA(c1, c2, c3, s5, c5);
    const unsigned long long c6 = s5 & c4;
    b0 = s4; b1 = s5 ^ c4; b2 = c5 ^ c6; b3 = c5 & c6;
}
__global__ __launch_bounds__(256) void k_thin_bits04(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, int H, int Ww, int sub, int* __restrict__ changed) {
    const size_t nw = (size_t)H * Ww, wi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (wi >= nw) return;
    const unsigned long long* s = src + nw * blockIdx.z; unsigned long long* d = dst + nw * blockIdx.z;
    const unsigned long long M = s[wi];
    if (!M) { d[wi] = 0; return; }
    const Nb8 n = neighbour_planes(s, H, Ww, (int)(wi / Ww), (int)(wi % Ww), M);
    unsigned long long b0, b1, b2, b3; count8(n, b0, b1, b2, b3);
    const unsigned long long Bok = (b1 | b2) & ~b3 & ~(b2 & b1 & b0);          // 2 <= B <= 6
    const unsigned long long P2 = n.S, P3 = n.SW, P4 = n.W, P5 = n.NW, P6 = n.N, P7 = n.NE, P8 = n.E, P9 = n.SE;
    unsigned long long one = 0, two = 0;
    auto TR = [&](unsigned long long a, unsigned long long b) { const unsigned long long t = ~a & b; two |= one & t; one |= t; };
    TR(P2, P3); TR(P3, P4); TR(P4, P5); TR(P5, P6); TR(P6, P7); TR(P7, P8); TR(P8, P9); TR(P9, P2);
    const unsigned long long cnd = sub == 0 ? (~(P2 & P4 & P6) & ~(P4 & P6 & P8)) : (~(P2 & P4 & P8) & ~(P2 & P6 & P8));
    const unsigned long long del = M & (one & ~two) & Bok & cnd;
    if (del) *changed = 1;
    d[wi] = M & ~del;
}
// skeleton bytes (0 / 255) and state bytes (ST_FG | ST_END for degree 1 | ST_JUN for degree >= 3) from the thinned bit planes
__global__ __launch_bounds__(256) void k_bits_to_skel_state(const unsigned long long* __restrict__ bits, u8* __restrict__ skel, u8* __restrict__ st, int H, int W, int Ww,
                                                            unsigned long long* __restrict__ d2bits /* optional: the degree-2 pixels as a bit plane (k_chain_ends_bits) */) {
    const size_t nw = (size_t)H * Ww, w0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (w0 >= nw) return;
    const unsigned long long* s = bits + nw * blockIdx.z;
    u8* sk = skel + (size_t)H * W * blockIdx.z; u8* so = st + (size_t)H * W * blockIdx.z;
    const int lane = threadIdx.x & 63;
    unsigned long long fg = 0, en = 0, ju = 0, d2 = 0;
    auto wi_of = [](size_t a, int l) { return a + (size_t)l; };
    if (w0 + lane < nw) {
        const size_t wi = w0 + lane;
        fg = s[wi];
        if (fg) {
            const Nb8 n = neighbour_planes(s, H, Ww, (int)(wi / Ww), (int)(wi % Ww), fg);
            unsigned long long b0, b1, b2, b3; count8(n, b0, b1, b2, b3);
            en = fg & b0 & ~b1 & ~b2 & ~b3;                   // exactly one neighbour
            ju = fg & (b2 | b3 | (b1 & b0));                  // three or more
            d2 = fg & ~b0 & b1 & ~b2 & ~b3;                   // exactly two: a walk passing through has no choice (walker.h: forced stretches)
        }
        if (d2bits) d2bits[nw * blockIdx.z + wi_of(w0, lane)] = d2;
    }
    if ((W & 63) == 0) {          // a word is 64 whole pixels of one row: every lane writes the 64 skeleton and 64 state bytes of its own word, 16 bytes per store
        if (w0 + lane >= nw) return;
        const size_t px = (w0 + lane) * 64;                 // == y * W + 64 xw
        uint4* ok = reinterpret_cast<uint4*>(sk + px); uint4* oo = reinterpret_cast<uint4*>(so + px);
        auto spread = [](unsigned long long v, int g) { return ((((unsigned)(v >> (4 * g)) & 0xfu) * 0x00204081u) & 0x01010101u); };      // 4 bits -> 0 / 1 in 4 bytes
#pragma unroll
        for (int q = 0; q < 4; q++) {
            unsigned a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int g = 4 * q + r;
                const unsigned F = spread(fg, g);
                a[r] = F * 0xffu;
                b[r] = (F << 7) | spread(en, g) | (spread(ju, g) << 1) | (spread(d2, g) << 2);      // ST_FG 0x80, ST_END 1, ST_JUN 2, ST_DEG2 4 (en, ju, d2 are subsets of fg)
            }
            ok[q] = make_uint4(a[0], a[1], a[2], a[3]); oo[q] = make_uint4(b[0], b[1], b[2], b[3]);
        }
        return;
    }
    auto bc = [&](unsigned long long v, int j) -> unsigned long long {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), j) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, j);
    };
    for (int j = 0; j < 64; j++) {
        const size_t wi = w0 + j; if (wi >= nw) break;
        const unsigned long long f = bc(fg, j), e = bc(en, j), q = bc(ju, j), t2 = bc(d2, j);
        const int y = (int)(wi / Ww), x = (int)(wi % Ww) * 64 + lane;
        if (x < W) {
            const bool on = (f >> lane) & 1ULL;
            sk[(size_t)y * W + x] = on ? 255 : 0;
            so[(size_t)y * W + x] = on ? (u8)(ST_FG | (((e >> lane) & 1ULL) ? ST_END : 0) | (((q >> lane) & 1ULL) ? ST_JUN : 0) | (((t2 >> lane) & 1ULL) ? ST_DEG2 : 0)) : (u8)0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Forced stretches (walker.h: ST_CHAIN): maximal chains of degree-2 skeleton pixels, listed when at least ORIP_CHAIN_MIN long.
//   k_chain_ends : a degree-2 pixel with a neighbour that is not degree-2 ends a chain
//   k_chain_build: one thread per end pixel walks its chain (each pixel has exactly one way on); the end with the smaller pixel index
//                  owns the chain, takes room in cpix with one atomic, walks it again and writes cpix / cref / the state flags
// Runs on lane 0's side stream underneath the component labelling; the traces wait for it.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chain_ends(const u8* __restrict__ st, int H, int W, int64_t n, unsigned* __restrict__ ends, unsigned* __restrict__ n_ends, unsigned cap) {
    const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    uint32_t w4 = 0;
    if (i4 + 3 < n) w4 = *reinterpret_cast<const uint32_t*>(st + i4); else for (int j = 0; j < 4 && i4 + j < n; j++) w4 |= (uint32_t)st[i4 + j] << (8 * j);
    if (!(w4 & 0x04040404u)) return;
    const int64_t plane = (int64_t)H * W;
    for (int j = 0; j < 4; j++) {
        if (!((w4 >> (8 * j)) & ST_DEG2)) continue;
        const int64_t g = i4 + j; const int layer = (int)(g / plane); const int64_t p = g - (int64_t)layer * plane;
        const int y = (int)(p / W), x = (int)(p % W);
        const u8* s = st + plane * layer;
        bool end = false;
        for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
            if (!dy && !dx) continue;
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const u8 v = s[(size_t)yy * W + xx];
            if ((v & ST_FG) && !(v & ST_DEG2)) end = true;
        }
        if (end) { const unsigned k = atomicAdd(n_ends, 1u); if (k < cap) ends[k] = ((unsigned)layer << 26) | (unsigned)p; }
    }
}
// the same from the bit planes: a word of 64 pixels at a time -- ends = degree-2 pixels with a skeleton neighbour that is not degree-2
__global__ __launch_bounds__(256) void k_chain_ends_bits(const unsigned long long* __restrict__ bits, const unsigned long long* __restrict__ d2bits, int H, int W, int Ww,
                                                          unsigned* __restrict__ ends, unsigned* __restrict__ n_ends, unsigned cap) {
    const size_t nw = (size_t)H * Ww, wi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (wi >= nw) return;
    const unsigned long long* d2p = d2bits + nw * blockIdx.z;
    const unsigned long long D = d2p[wi];
    if (!D) return;
    const unsigned long long* fgp = bits + nw * blockIdx.z;
    const int y = (int)(wi / Ww), xw = (int)(wi % Ww);
    auto ND = [&](int yy, int xx) -> unsigned long long { return (yy < 0 || yy >= H || xx < 0 || xx >= Ww) ? 0ULL : (fgp[(size_t)yy * Ww + xx] & ~d2p[(size_t)yy * Ww + xx]); };      // skeleton, not degree 2
    const unsigned long long U = ND(y - 1, xw), UL = ND(y - 1, xw - 1), UR = ND(y - 1, xw + 1), M = ND(y, xw), ML = ND(y, xw - 1), MR = ND(y, xw + 1);
    const unsigned long long Dn = ND(y + 1, xw), DL = ND(y + 1, xw - 1), DR = ND(y + 1, xw + 1);
    const unsigned long long any = U | ((U >> 1) | (UR << 63)) | ((U << 1) | (UL >> 63)) | ((M >> 1) | (MR << 63)) | ((M << 1) | (ML >> 63)) | Dn | ((Dn >> 1) | (DR << 63)) | ((Dn << 1) | (DL >> 63));
    unsigned long long e = D & any;
    while (e) {
        const int j = __ffsll((long long)e) - 1; e &= e - 1;
        const int x = xw * 64 + j;
        if (x >= W) break;
        const unsigned k = atomicAdd(n_ends, 1u);
        if (k < cap) ends[k] = ((unsigned)blockIdx.z << 26) | ((unsigned)y * (unsigned)W + (unsigned)x);
    }
}
__global__ __launch_bounds__(64) void k_chain_build(u8* __restrict__ st, int H, int W, const unsigned* __restrict__ ends, const unsigned* __restrict__ n_ends, unsigned cap_ends,
                                                     unsigned* __restrict__ cpix, unsigned* __restrict__ cref, unsigned* __restrict__ n_cpix, unsigned cap_cpix) {
    const unsigned ne = min(*n_ends, cap_ends);
    const unsigned e = blockIdx.x * 64 + threadIdx.x;
    if (e >= ne) return;
    const int layer = (int)(ends[e] >> 26); const unsigned p0 = ends[e] & 0x3ffffffu;
    const int64_t plane = (int64_t)H * W;
    u8* s = st + plane * layer; unsigned* cr = cref + plane * layer;
    // the way on from `cur` (a degree-2 pixel): its degree-2 neighbour that is not `prev`; ~0u: the chain ends here
    auto next_of = [&](unsigned cur, unsigned prev) -> unsigned {
        const int y = (int)(cur / (unsigned)W), x = (int)(cur % (unsigned)W);
        for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
            if (!dy && !dx) continue;
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const unsigned q = (unsigned)yy * (unsigned)W + (unsigned)xx;
            if (q == prev) continue;
            const u8 v = s[q];
            if ((v & ST_FG) && (v & ST_DEG2)) return q;
        }
        return ~0u;
    };
    unsigned m = 1, cur = p0, prev = ~0u;
    for (;;) { const unsigned nx = next_of(cur, prev); if (nx == ~0u || m > (1u << 24)) break; prev = cur; cur = nx; m++; }
    if (m < ORIP_CHAIN_MIN || !(p0 < cur)) return;                   // short, or the other end owns it
    const unsigned base = atomicAdd(n_cpix, m + 2u);
    if (base + m + 2u > cap_cpix) return;
    cpix[base] = ORIP_CHAIN_SENTINEL; cpix[base + m + 1u] = ORIP_CHAIN_SENTINEL;
    cur = p0; prev = ~0u;
    for (unsigned j = 0; j < m; j++) {
        cpix[base + 1u + j] = cur; cr[cur] = base + 1u + j;
        s[cur] = (u8)(s[cur] | ST_CHAIN | ((j == 0 || j == m - 1u) ? ST_CHAIN_END : 0));
        const unsigned nx = next_of(cur, prev); prev = cur; cur = nx;
    }
}

// State kept between orip_contours_prepare and the per-layer trace calls (device buffers live in lane 0's scratch).
struct Prep04 {
    bool ready = false;
    unsigned M = 0, NC = 0;
    int K = 0;                                      // layer count the schedule was built for (c->K may change afterwards)
    std::vector<unsigned> h_cs, layer_first;       // comp_start on the host; first component of every layer (+ sentinel)
    const unsigned* order = nullptr;                // components by (layer, size descending)
    WalkArgs A;                                     // shared arguments (state bytes, keys, lin, comp_start, memo, winfo, total_fg)
    unsigned F[ORIP_MAX_LAYERS];                    // log capacity factor of the trace in flight
    bool launched[ORIP_MAX_LAYERS];
    bool memo_clear[ORIP_MAX_LAYERS];               // the layer's memo plane was zeroed on its lane while the raster part ran
    bool chains = false;                            // forced stretches are listed (cref / cpix) and flagged in the state plane
};
void orip_contours_free(orip_ctx* c) { delete static_cast<Prep04*>(c->prep04); c->prep04 = nullptr; }

__global__ __launch_bounds__(256) void k_comp_order_keys(const unsigned* __restrict__ keys, const unsigned* __restrict__ cs, unsigned nc, unsigned long long* __restrict__ k, unsigned* __restrict__ idx) {
    unsigned c = blockIdx.x * 256 + threadIdx.x;
    if (c < nc) { k[c] = ((unsigned long long)(keys[cs[c]] >> 26) << 32) | (unsigned)~(cs[c + 1] - cs[c]); idx[c] = c; }
}
__global__ __launch_bounds__(256) void k_clear_visited_layer(u8* __restrict__ st, const unsigned* __restrict__ lin, int64_t m) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < m) st[lin[i]] &= (u8)~ST_VIS;
}
__global__ __launch_bounds__(256) void k_kept_slots_base(const unsigned* __restrict__ kept, const unsigned* __restrict__ path_off, const unsigned long long* __restrict__ pts_off, unsigned n, unsigned base,
                                                         unsigned* __restrict__ slots, unsigned long long* __restrict__ slot_off) {
    unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && kept[i]) { slots[path_off[i]] = base + i; slot_off[path_off[i]] = pts_off[i]; }
}

// Everything of stage 04 that is batched over the layers: thinning, components, state bytes, the raster-ordered pixel list
// sorted by component, the per-layer schedule.  Runs on lane 0 and ends synchronised.
// Hint for the resident chain: the image is set and K layers will be traced.  Clears the K memo planes on the layer lanes NOW -- at the start of a
// step the card is idle but for the k-means fit, whereas 4 GB of fills issued from orip_contours_prepare run into stages 02 / 03 -- and
// orip_contours_prepare then skips its own clearing.  Nothing else touches lane 0's vtmp[6] in between.
extern "C" int orip_contours_reserve(orip_ctx* c, int K) {
    orip_enter(c);
    if (!c->image.p || c->H <= 0 || c->W <= 0) ORIP_FAIL(c, "no image set");
    if (K < 1 || K > ORIP_MAX_LAYERS) ORIP_FAIL(c, "K=%d out of range 1..%d", K, ORIP_MAX_LAYERS);
    const size_t plane = (size_t)c->H * c->W;
    c->memo_pre_K = 0;
    HIPC(c, LN(c).vtmp[6].ensure(plane * (size_t)K * 8 * 4 + 64));
    for (int l = 0; l < K; l++) {
        ORIP_LANE(c, l + 1);
        HIPC(c, hipMemsetAsync(c->ln[0].vtmp[6].as<unsigned>() + plane * 8 * l, 0, plane * 8 * 4, LN(c).stream));
    }
    c->memo_pre_K = K; c->memo_pre_H = c->H; c->memo_pre_W = c->W;
    return 0;
}

static int trace_launch(orip_ctx* c, Prep04& R, int layer, unsigned F);
extern "C" int orip_contours_prepare(orip_ctx* c) {
    orip_enter(c);
    if (!c->edges.p || c->K < 1) ORIP_FAIL(c, "no edges resident (run orip_detect_edges or orip_set_edges)");
    const int H = c->H, W = c->W, K = c->K;
    if (H > 8192 || W > 8192) ORIP_FAIL(c, "image %dx%d exceeds the 8192x8192 limit of the component key packing", W, H);
    if (!c->prep04) c->prep04 = new Prep04();
    Prep04& R = *static_cast<Prep04*>(c->prep04);
    R.ready = false; R.K = K;
    for (int l = 0; l < ORIP_MAX_LAYERS; l++) { R.launched[l] = false; R.F[l] = 0; R.memo_clear[l] = false; }
    const size_t plane = (size_t)H * W; const int64_t n = (int64_t)plane * K;
    // the memo planes (one word per pixel and incoming direction, 0.5 GB per layer at 4096^2) are cleared on the layer lanes, not in front of
    // every layer's trace: by orip_contours_reserve at the start of the step when the caller gave that hint, else now, underneath the raster work below
    const bool pre = c->memo_pre_K >= K && c->memo_pre_H == H && c->memo_pre_W == W;
    c->memo_pre_K = 0;
    HIPC(c, LN(c).vtmp[6].ensure(plane * (size_t)K * 8 * 4 + 64));
    for (int l = 0; l < K; l++) {
        if (!pre) {
            ORIP_LANE(c, l + 1);
            HIPC(c, hipMemsetAsync(c->ln[0].vtmp[6].as<unsigned>() + plane * 8 * l, 0, plane * 8 * 4, LN(c).stream));
        }
        R.memo_clear[l] = true;
    }
    // ---- thinning_zhangsuen (04:35-99): <=120 iterations of two sub-iterations, until nothing is deleted
    HIPC(c, c->skel.ensure(plane * K + 16));
    HIPC(c, c->tmpB.ensure(plane * K + 16));
    HIPC(c, LN(c).flags.ensure(1024));
    int* d_changed = LN(c).flags.as<int>() + 8; (void)d_changed;
    dim3 g2(cdiv(W, 64), cdiv(H, 4), K), block(256);
    const int Wb = (W + 1) >> 1, Hb = (H + 1) >> 1; const size_t pplane = (size_t)Wb * Hb * 4;
    HIPC(c, c->tmpC.ensure(plane * K + 16));   // state bytes
    const unsigned long long* d2_plane = nullptr;   // degree-2 pixels as a bit plane (bit-plane thinning path only)
    if (!ORIP_VARIANT("ORIP_THIN_BYTES")) {
        // bit planes: 2 MB per 4096^2 layer; pack, iterate, then skeleton and state bytes in one unpacking pass
        const int Ww = (W + 63) >> 6; const size_t nw = (size_t)H * Ww;
        HIPC(c, LN(c).vtmp[10].ensure(nw * K * 16 + 64));
        unsigned long long* bA = LN(c).vtmp[10].as<unsigned long long>(); unsigned long long* bB = bA + nw * K;
        dim3 gw((unsigned)cdiv((int64_t)nw, 256), 1, K);
        if (c->edge_bits != (const void*)bA) hipLaunchKernelGGL(k_bytes_to_bits04, gw, block, 0, LN(c).stream, c->edges.as<u8>(), bA, H, W, Ww);   // stage 03 may have left them
        c->edge_bits = nullptr;
        // two iterations per round trip to the host, each with its own flag (an iteration after an unchanged one changes nothing either)
        int* d_ch2 = LN(c).flags.as<int>() + 212;
        for (int it = 0; it < 120; it += 2) {
            HIPC(c, hipMemsetAsync(d_ch2, 0, 8, LN(c).stream));
            for (int b = 0; b < 2; b++) {
                { ProfScope ps(c, "k_thin_bits"); hipLaunchKernelGGL(k_thin_bits04, gw, block, 0, LN(c).stream, bA, bB, H, Ww, 0, d_ch2 + b); }
                { ProfScope ps(c, "k_thin_bits"); hipLaunchKernelGGL(k_thin_bits04, gw, block, 0, LN(c).stream, bB, bA, H, Ww, 1, d_ch2 + b); }
            }
            int h_changed[2] = {0, 0};
            HIPC(c, hipMemcpyAsync(h_changed, d_ch2, 8, hipMemcpyDeviceToHost, LN(c).stream));
            HIPC(c, hipStreamSynchronize(LN(c).stream));
            if (!(h_changed[0] && h_changed[1])) break;
        }
        { ProfScope ps(c, "k_skel_state"); hipLaunchKernelGGL(k_bits_to_skel_state, gw, block, 0, LN(c).stream, bA, c->skel.as<u8>(), c->tmpC.as<u8>(), H, W, Ww, bB); }
        d2_plane = bB;                                   // (the second thinning plane is free now)
    }
#ifdef ORIP_VARIANTS
    else {
        // iteration 1 reads the edges; ping-pong skel <-> tmpB so that the result always lands in skel
        const u8* cur = c->edges.as<u8>();
        for (int it = 0; it < 120; it++) {
            HIPC(c, hipMemsetAsync(d_changed, 0, 4, LN(c).stream));
            { ProfScope ps(c, "k_thin_sub"); hipLaunchKernelGGL(k_thin_sub, g2, block, 0, LN(c).stream, cur, c->tmpB.as<u8>(), H, W, 0, d_changed); }
            { ProfScope ps(c, "k_thin_sub"); hipLaunchKernelGGL(k_thin_sub, g2, block, 0, LN(c).stream, c->tmpB.as<u8>(), c->skel.as<u8>(), H, W, 1, d_changed); }
            cur = c->skel.as<u8>();
            int h_changed = 0;
            HIPC(c, hipMemcpyAsync(&h_changed, d_changed, 4, hipMemcpyDeviceToHost, LN(c).stream));
            HIPC(c, hipStreamSynchronize(LN(c).stream));
            if (!h_changed) break;
        }
        { ProfScope ps(c, "k_skel_state"); hipLaunchKernelGGL(k_skel_state, g2, block, 0, LN(c).stream, c->skel.as<u8>(), c->tmpC.as<u8>(), H, W); }
    }
#endif
    // ---- forced stretches (walker.h: ST_CHAIN), on the side stream underneath the component work below; the traces wait for ev3.
    // Sized from the skeleton of the previous prepare of this context (a resident chain repeats itself), else from the pixel count: chains
    // that do not fit are simply not listed (k_chain_build), the walker then steps through them as before.
    R.chains = !getenv("ORIP_NO_CHAINS");
    if (R.chains) {
        const size_t m_guess = R.M ? (size_t)R.M + R.M / 4 + 4096 : (size_t)(n / 16 + 4096);
        const unsigned cap_ends = (unsigned)std::min<size_t>(m_guess, 0x3fffffffu), cap_cpix = (unsigned)std::min<size_t>(2 * m_guess + 256, 0x7fffffffu);
        HIPC(c, c->cref.ensure(plane * (size_t)K * 4 + 64));
        HIPC(c, c->cpix.ensure((size_t)cap_cpix * 4 + (size_t)cap_ends * 4 + 64));
        unsigned* cpix = c->cpix.as<unsigned>(); unsigned* ends = cpix + cap_cpix;
        unsigned* d_cn = LN(c).flags.as<unsigned>() + 232;                          // {ends, cpix entries}
        hipStream_t s2 = LN(c).stream2;
        HIPC(c, hipEventRecord(LN(c).ev2, LN(c).stream));                            // the state bytes
        HIPC(c, hipStreamWaitEvent(s2, LN(c).ev2, 0));
        HIPC(c, hipMemsetAsync(cpix, 0xff, (size_t)cap_cpix * 4, s2));             // sentinels everywhere, 64 of them in front of the first chain
        const unsigned init[2] = {0u, 64u};
        HIPC(c, hipMemcpyAsync(d_cn, init, 8, hipMemcpyHostToDevice, s2));
        if (d2_plane) {
            const int Ww = (W + 63) >> 6;
            hipLaunchKernelGGL(k_chain_ends_bits, dim3((unsigned)cdiv((int64_t)H * Ww, 256), 1, K), block, 0, s2, LN(c).vtmp[10].as<unsigned long long>(), d2_plane, H, W, Ww, ends, d_cn, cap_ends);
        } else hipLaunchKernelGGL(k_chain_ends, dim3(cdiv(cdiv(n, 4), 256)), block, 0, s2, c->tmpC.as<u8>(), H, W, n, ends, d_cn, cap_ends);
        hipLaunchKernelGGL(k_chain_build, dim3(cdiv(cap_ends, 64)), dim3(64), 0, s2, c->tmpC.as<u8>(), H, W, ends, d_cn, cap_ends, cpix, c->cref.as<unsigned>(), d_cn + 1, cap_cpix - 64u);
        HIPC(c, hipEventRecord(LN(c).ev3, s2));
    }
    // ---- components (from the thinned bit planes when they exist)
    HIPC(c, c->tmpD.ensure(pplane * K * sizeof(int)));
    if (!ORIP_VARIANT("ORIP_THIN_BYTES") && !getenv("ORIP_CCL_BYTES")) ORIP_TRY(orip_ccl_bits(c, LN(c).vtmp[10].as<unsigned long long>(), c->tmpD.as<int>(), K));
    else ORIP_TRY(orip_ccl(c, c->skel.as<u8>(), c->tmpD.as<int>(), K, 0));
    // ---- ordered compaction
    const bool from_bits = !ORIP_VARIANT("ORIP_THIN_BYTES") && !getenv("ORIP_COMPACT_BYTES");       // the thinned bit planes are in vtmp[10]
    const int Wwc = (W + 63) >> 6; const size_t nwc = (size_t)H * Wwc, nwords = nwc * K;
    const unsigned long long* sk_bits = LN(c).vtmp[10].as<unsigned long long>();
    const int nblk = from_bits ? (int)cdiv((int64_t)nwords, 256) : cdiv(n, 1024);
    HIPC(c, LN(c).tmpE.ensure((size_t)(nblk + 1) * 2 * sizeof(unsigned) + 64));
    unsigned* d_cnt = LN(c).tmpE.as<unsigned>(); unsigned* d_boff = d_cnt + nblk + 1;
    HIPC(c, hipMemsetAsync(d_cnt + nblk, 0, sizeof(unsigned), LN(c).stream));
    { ProfScope ps(c, "k_compact_count");
      if (from_bits) hipLaunchKernelGGL(k_compact_count_bits, dim3(nblk), block, 0, LN(c).stream, sk_bits, nwords, d_cnt);
      else hipLaunchKernelGGL(k_compact_count, dim3(nblk), block, 0, LN(c).stream, c->tmpC.as<u8>(), n, d_cnt); }
    ORIP_TRY(excl_scan<unsigned>(c, d_cnt, d_boff, (size_t)nblk + 1, LN(c).tmpF));
    unsigned M = 0;
    HIPC(c, hipMemcpyAsync(&M, d_boff + nblk, sizeof(unsigned), hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    for (int l = 0; l < K; l++) {
        DPolys& P = c->polys[ORIP_SLOT_CONTOURS][l];
        P.n = 0; P.total = 0; P.set_explicit();
        HIPC(c, P.off.ensure(8)); HIPC(c, hipMemsetAsync(P.off.p, 0, 8, LN(c).stream));
    }
    R.M = M; R.NC = 0;
    if (M == 0) { HIPC(c, hipStreamSynchronize(LN(c).stream)); R.ready = true; return 0; }
    // keys / lin (double buffers for the sort)
    HIPC(c, LN(c).vtmp[0].ensure((size_t)M * 4 * 4 + 64));
    unsigned* keys_in = LN(c).vtmp[0].as<unsigned>(); unsigned* lin_in = keys_in + M; unsigned* keys = lin_in + M; unsigned* lin = keys + M;
    { ProfScope ps(c, "k_compact_write");
      if (from_bits) hipLaunchKernelGGL(k_compact_write_bits, dim3(nblk), block, 0, LN(c).stream, sk_bits, c->tmpD.as<int>(), nwords, nwc, H, W, Wwc, d_boff, keys_in, lin_in);
      else hipLaunchKernelGGL(k_compact_write, dim3(nblk), block, 0, LN(c).stream, c->tmpC.as<u8>(), c->tmpD.as<int>(), n, H, W, d_boff, keys_in, lin_in); }
    {
        size_t bytes = 0;
        HIPC(c, rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys, lin_in, lin, (size_t)M, 0, 30, LN(c).stream));
        HIPC(c, LN(c).tmpF.ensure(bytes + 16));
        ProfScope ps(c, "radix_sort_pairs");
        HIPC(c, rocprim::radix_sort_pairs(LN(c).tmpF.p, bytes, keys_in, keys, lin_in, lin, (size_t)M, 0, 30, LN(c).stream));
    }
    // ---- component segmentation
    HIPC(c, LN(c).vtmp[1].ensure((size_t)M * 2 * 4 + 64));
    unsigned* head = LN(c).vtmp[1].as<unsigned>(); unsigned* head_scan = head + M;
    hipLaunchKernelGGL(k_heads, dim3(cdiv(M, 256)), block, 0, LN(c).stream, keys, (int64_t)M, head);
    ORIP_TRY(excl_scan<unsigned>(c, head, head_scan, (size_t)M, LN(c).tmpF));
    unsigned last2[2];
    HIPC(c, hipMemcpyAsync(&last2[0], head_scan + (M - 1), 4, hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipMemcpyAsync(&last2[1], head + (M - 1), 4, hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    const unsigned NC = last2[0] + last2[1];
    R.NC = NC;
    HIPC(c, LN(c).vtmp[2].ensure((size_t)(NC + 2) * 4 + 64));
    unsigned* comp_start = LN(c).vtmp[2].as<unsigned>();
    hipLaunchKernelGGL(k_comp_starts, dim3(cdiv(M, 256)), block, 0, LN(c).stream, head, head_scan, (int64_t)M, comp_start, NC);
    R.h_cs.assign(NC + 1, 0); std::vector<unsigned> h_keyfirst(NC);
    HIPC(c, hipMemcpyAsync(R.h_cs.data(), comp_start, (size_t)(NC + 1) * 4, hipMemcpyDeviceToHost, LN(c).stream));
    // layer of each component (key of its first element)
    HIPC(c, LN(c).vtmp[3].ensure((size_t)NC * 4 + 64));
    WalkArgs& A = R.A; memset(&A, 0, sizeof(A));
    A.H = H; A.W = W; A.plane = (int64_t)plane; A.st = c->tmpC.as<u8>(); A.keys = keys; A.lin = lin; A.comp_start = comp_start; A.nc = NC;
    if (R.chains) { A.cref = c->cref.as<unsigned>(); A.cpix = c->cpix.as<unsigned>(); }
    hipLaunchKernelGGL(k_gather_head_layers, dim3(cdiv(NC, 256)), block, 0, LN(c).stream, keys, comp_start, NC, LN(c).vtmp[3].as<unsigned>());
    HIPC(c, hipMemcpyAsync(h_keyfirst.data(), LN(c).vtmp[3].p, (size_t)NC * 4, hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    std::vector<unsigned>& layer_first = R.layer_first; layer_first.assign(K + 1, NC);
    for (unsigned i = NC; i-- > 0;) layer_first[h_keyfirst[i]] = i;
    for (int l = K - 1; l >= 0; l--) if (layer_first[l] == NC && l + 1 <= K) layer_first[l] = layer_first[l + 1];
    layer_first[K] = NC;
    for (int l = 0; l < K; l++) A.total_fg[l] = (long long)R.h_cs[layer_first[l + 1]] - (long long)R.h_cs[layer_first[l]];
    // per-layer largest-first schedule: components sorted by (layer, size descending); layer l owns order[layer_first[l] .. layer_first[l+1])
    {
        HIPC(c, LN(c).vtmp[5].ensure((size_t)NC * 24 + 64));
        unsigned long long* kin = LN(c).vtmp[5].as<unsigned long long>(); unsigned long long* kout = kin + NC; unsigned* idin = (unsigned*)(kout + NC); unsigned* idout = idin + NC;
        hipLaunchKernelGGL(k_comp_order_keys, dim3(cdiv(NC, 256)), block, 0, LN(c).stream, keys, comp_start, NC, kin, idin);
        size_t bytes = 0;
        HIPC(c, rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, idin, idout, (size_t)NC, 0, 40, LN(c).stream));
        HIPC(c, LN(c).tmpF.ensure(bytes + 16));
        HIPC(c, rocprim::radix_sort_pairs(LN(c).tmpF.p, bytes, kin, kout, idin, idout, (size_t)NC, 0, 40, LN(c).stream));
        HIPC(c, LN(c).vtmp[3].ensure((size_t)NC * 4 + 64));
        HIPC(c, hipMemcpyAsync(LN(c).vtmp[3].p, idout, (size_t)NC * 4, hipMemcpyDeviceToDevice, LN(c).stream));
        R.order = LN(c).vtmp[3].as<unsigned>();
    }
    // shared trace state: memo plane (one word per pixel and incoming direction) and walk records (two slots per skeleton pixel)
    HIPC(c, LN(c).vtmp[6].ensure(plane * (size_t)K * 8 * 4 + 64));
    HIPC(c, LN(c).vtmp[8].ensure((size_t)2 * M * sizeof(WalkInfo) + 64));
    A.memo = LN(c).vtmp[6].as<unsigned>(); A.winfo = LN(c).vtmp[8].as<WalkInfo>();
    HIPC(c, LN(c).vtmp[7].ensure((size_t)(NC + 1) * 4 + 64));
    A.log_used = LN(c).vtmp[7].as<unsigned>();         // written by every component's trace
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    R.ready = true;
    // The layers' traces start right here, from the calling thread, each on its layer's lane: the walks head every layer's chain, and the hand-over to the
    // layer threads (orip_contours_layer, which then only waits for its trace) costs 0.3-0.5 ms in which the card would sit idle.  A lane that another
    // thread holds is left to its orip_contours_layer call (ORIP_TRACE_LATE=1: all of them, as before).
    if (!getenv("ORIP_TRACE_LATE"))
        for (int l = 0; l < R.K; l++) {
            if (R.layer_first[l] == R.layer_first[l + 1]) continue;
            LaneGuard g(c, l + 1);
            if (!g.ok) continue;
            HIPC(c, orip_pf08_drain(c));
            ORIP_TRY(trace_launch(c, R, l, 64));
        }
    return 0;
}

// Enqueues the trace pass of one layer on the calling lane's stream (walker.h: one wave per component records step codes, bounce
// trajectories and one WalkInfo per walk).  The logs of the layer live in the lane's scratch; the kernel addresses them with the
// global formula (cap_factor * comp_start + ...), so the base pointers are shifted by the layer's first component.
static int trace_launch(orip_ctx* c, Prep04& R, int layer, unsigned F) {
    const unsigned c0 = R.layer_first[layer], c1 = R.layer_first[layer + 1];
    const unsigned b0 = R.h_cs[c0], b1 = R.h_cs[c1];
    const unsigned Ml = b1 - b0, NCl = c1 - c0;
    const size_t plane = (size_t)R.A.plane;
    if ((uint64_t)F * R.M + (uint64_t)256 * R.NC + 64 >= 0xffffffffull) ORIP_FAIL(c, "skeleton too large for the walk logs (factor %u)", F);
    const size_t nlog = (size_t)F * Ml + (size_t)64 * NCl + 8, nstep = (size_t)F * Ml + (size_t)256 * NCl + 8;
    WalkStore& WS = c->wstore[layer];
    WS.epoch++; WS.n = 0;                              // walk-coded lists built on the previous trace of this layer are stale from here on
    HIPC(c, WS.log.ensure(nlog * 16 + 64));
    HIPC(c, LN(c).vtmp[9].ensure(nstep + 64));
    WalkArgs A = R.A;
    A.logbuf = WS.log.as<unsigned>() - 4 * ((size_t)F * b0 + (size_t)64 * c0);
    A.steplog = LN(c).vtmp[9].as<u8>() - ((size_t)F * b0 + (size_t)256 * c0);
    A.cap_factor = F; A.comp_order = R.order + c0; A.nc = NCl;
    int* d_over = LN(c).flags.as<int>() + 20; A.overflow = d_over;
    if (R.chains) HIPC(c, hipStreamWaitEvent(LN(c).stream, c->ln[0].ev3, 0));            // the chain lists and flags (orip_contours_prepare, side stream)
    if (!R.memo_clear[layer]) HIPC(c, hipMemsetAsync(A.memo + plane * 8 * layer, 0, plane * 8 * 4, LN(c).stream));
    R.memo_clear[layer] = false;                     // a retry (or a second trace without prepare) clears it itself
    HIPC(c, hipMemsetAsync(A.winfo + 2 * (size_t)b0, 0, (size_t)2 * Ml * sizeof(WalkInfo), LN(c).stream));
    HIPC(c, hipMemsetAsync(d_over, 0, 4, LN(c).stream));
    if (getenv("ORIP_WALK_DBG")) {          // per-component counters (walks, steps, memo hits, closed cycles, tile loads, size)
        HIPC(c, LN(c).vtmp[11].ensure((size_t)NCl * 256 + 64)); HIPC(c, hipMemsetAsync(LN(c).vtmp[11].p, 0, (size_t)NCl * 256, LN(c).stream));
        A.dbg = LN(c).vtmp[11].as<unsigned long long>() - 32ull * c0;
    }
    { ProfScope ps(c, "k_trace"); hipLaunchKernelGGL(k_trace, dim3(NCl), dim3(64), 0, LN(c).stream, A); }
    HIPC(c, hipGetLastError());
    R.F[layer] = F; R.launched[layer] = true;
    return 0;
}
// Waits for the trace of the layer (retrying with larger logs on overflow), then sizes, allocates and writes its contours.
static int trace_finish(orip_ctx* c, Prep04& R, int layer) {
    const unsigned c0 = R.layer_first[layer], c1 = R.layer_first[layer + 1];
    const unsigned b0 = R.h_cs[c0], b1 = R.h_cs[c1];
    const unsigned Ml = b1 - b0;
    const size_t plane = (size_t)R.A.plane;
    dim3 block(256);
    int* d_over = LN(c).flags.as<int>() + 20;
    // offsets: exclusive scans over the layer's walk slots (slot order == output order: components by rank, endpoint walks then leftovers, each in raster order)
    // Nothing waits for the trace first: the sizing kernels are enqueued behind it and the one read below brings its overflow flag along
    // with the totals (an overflowed trace -- rare: the logs hold 64 entries per skeleton pixel -- is redone with larger logs).
    const unsigned nslots = 2u * Ml, sl0 = 2u * b0;
    const size_t ns1 = (size_t)nslots + 1;
    HIPC(c, LN(c).vtmp[4].ensure(ns1 * 2 * sizeof(WSum) + 256));
    WSum* ws = LN(c).vtmp[4].as<WSum>(); WSum* wo = ws + ns1;
    WalkStore& WS = c->wstore[layer];
    WSum h_tot; unsigned log_shift = 0; WalkArgs A;
    for (;;) {
    log_shift = (unsigned)((size_t)R.F[layer] * b0 + (size_t)64 * c0);
    A = R.A;                        // the layer's logs as the trace addressed them (walk_close_tail / vwalk_pieces follow the recorded trajectories)
    A.logbuf = WS.log.as<unsigned>() - 4 * (size_t)log_shift;
    A.steplog = LN(c).vtmp[9].as<u8>() - ((size_t)R.F[layer] * b0 + (size_t)256 * c0);
    A.cap_factor = R.F[layer]; A.overflow = d_over;
    hipLaunchKernelGGL(k_winfo_lens, dim3(cdiv(nslots + 1, 256)), block, 0, LN(c).stream, A, sl0, nslots, ws);
    {
        size_t bytes = 0; WSum zero; zero.pts = 0; zero.paths = zero.own = zero.pieces = zero.pad = 0;
        HIPC(c, rocprim::exclusive_scan(nullptr, bytes, ws, wo, zero, ns1, rocprim::plus<WSum>(), LN(c).stream));
        HIPC(c, LN(c).tmpF.ensure(bytes + 16));
        HIPC(c, rocprim::exclusive_scan(LN(c).tmpF.p, bytes, ws, wo, zero, ns1, rocprim::plus<WSum>(), LN(c).stream));
    }
    A.overflow = d_over;
    hipLaunchKernelGGL(k_totals_flag, dim3(1), dim3(1), 0, LN(c).stream, d_over, &wo[nslots].pad);
    HIPC(c, hipMemcpyAsync(&h_tot, wo + nslots, sizeof(WSum), hipMemcpyDeviceToHost, LN(c).stream));
    // pixels of the log entries in use (needs nothing from the scan: it runs while the host waits for the totals)
    {
        const unsigned NCl = c1 - c0;
        WS.ent_cap = (int64_t)8 * Ml + 64;                       // a component logs at most one entry per (pixel, incoming direction)
        HIPC(c, WS.lxy.ensure(((size_t)R.F[layer] * Ml + (size_t)64 * NCl + 8) * 8 + 64));
        HIPC(c, WS.ent_idx.ensure((size_t)WS.ent_cap * 4 + 64));
        HIPC(c, WS.cnt.ensure(64));
        HIPC(c, hipMemsetAsync(WS.cnt.p, 0, 4, LN(c).stream));
        hipLaunchKernelGGL(k_ent_fill, dim3(NCl, 8), block, 0, LN(c).stream, A, c0, log_shift, WS.lxy.as<int2>(), WS.ent_idx.as<unsigned>(), WS.cnt.as<unsigned>(), (unsigned)WS.ent_cap);
    }
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    if (!h_tot.pad) break;
    if (h_tot.pad == 2) ORIP_FAIL(c, "internal error: the walker of layer %d stopped making progress", layer);
    hipLaunchKernelGGL(k_clear_visited_layer, dim3(cdiv(Ml, 256)), block, 0, LN(c).stream, R.A.st + plane * layer, R.A.lin + b0, (int64_t)Ml);   // retry with larger logs
    ORIP_TRY(trace_launch(c, R, layer, R.F[layer] * 4));
    }
    if (getenv("ORIP_WALK_DBG")) {
        const unsigned NCl = c1 - c0;
        std::vector<unsigned long long> h((size_t)NCl * 32);
        HIPC(c, hipMemcpy(h.data(), LN(c).vtmp[11].p, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long tot[32] = {0}; size_t big = 0;
        for (size_t i = 0; i < h.size(); i++) tot[i % 32] += h[i];
        for (size_t i = 0; i < NCl; i++) if (h[i * 32 + 7] > h[big * 32 + 7]) big = i;
        const unsigned long long* d = &h[big * 32];
        fprintf(stderr, "[walk dbg] layer %d NC=%u M=%u F=%u: w1=%llu s1=%llu w2=%llu s2=%llu hit=%llu det=%llu tiles=%llu | largest fg=%llu: w1=%llu s1=%llu w2=%llu s2=%llu hit=%llu det=%llu tiles=%llu jumped=%llu\n",
                layer, NCl, Ml, R.F[layer], tot[0], tot[1], tot[2], tot[3], tot[4], tot[5], tot[6], d[7], d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[15]);
        if (d[8]) fprintf(stderr, "[walk prof] layer %d largest component: cycles total=%llu window loads=%llu scans=%llu look-ups=%llu | look-ups=%llu loop exits=%llu\n",
                          layer, d[8], d[9], d[10], d[11], d[12], d[13]);
        if (d[8]) fprintf(stderr, "[walk prof]   inside the look-ups: duplicate check %llu cycles (exact check in %llu look-ups); forced stretches %llu cycles in %llu calls, %llu rounds; stepping loop %llu cycles; walk start + end %llu cycles\n", d[14], d[20], d[16], d[17], d[18], d[19], d[21]);
    }
    const unsigned long long h_pts = h_tot.pts; const unsigned h_paths = h_tot.paths, h_own = h_tot.own, h_pieces = h_tot.pieces;
    // ---- the contours of the layer in walk-coded form (walker.h): nothing is expanded here
    DPolys& P = c->polys[ORIP_SLOT_CONTOURS][layer];
    P.total = (int64_t)h_pts; P.n = (int64_t)h_paths;
    P.virt = true; P.pts_ok = false; P.vident = true; P.vlayer = layer; P.vepoch = WS.epoch; P.scaled = false;
    HIPC(c, P.off.ensure((size_t)(P.n + 1) * 8 + 64));
    HIPC(c, WS.walk.ensure((size_t)std::max(h_paths, 1u) * sizeof(VWalk) + 64));
    HIPC(c, WS.piece.ensure((size_t)std::max(h_pieces, 1u) * sizeof(VPiece) + 64));
    HIPC(c, WS.own.ensure((size_t)std::max(h_own, 1u) * 8 + 64));
    HIPC(c, LN(c).vtmp[5].ensure((size_t)std::max(h_paths, 1u) * 4 + 64));
    unsigned* kept_slots = LN(c).vtmp[5].as<unsigned>();
    WS.n = P.n; WS.W = R.A.W; WS.n_own = h_own;
    hipLaunchKernelGGL(k_vwalk_fill, dim3(cdiv(nslots + 1, 256)), block, 0, LN(c).stream, A, sl0, nslots, wo, log_shift,
                       WS.walk.as<VWalk>(), WS.piece.as<VPiece>(), kept_slots, P.off.as<int64_t>());
    if (h_paths) {
        ProfScope ps(c, "k_write_walks");
        const unsigned chunk = 2048;
        hipLaunchKernelGGL(k_vown, dim3((unsigned)std::min<unsigned long long>(((unsigned long long)h_own + chunk - 1) / chunk, 262144ull)), dim3(64), 0, LN(c).stream, A, kept_slots, WS.walk.as<VWalk>(), h_paths,
                           WS.own.as<int2>(), h_own, chunk);
    }
    HIPC(c, hipGetLastError());
    R.launched[layer] = false;
    return 0;
}

// Contours of one layer (after orip_contours_prepare).  Runs on the layer's own lane, so different layers can be traced from
// different host threads at the same time and a finished layer can move on to stages 05-08 while others are still walking.
int orip_contours_layer_impl(orip_ctx* c, int layer, bool sync) {
    Prep04* R = static_cast<Prep04*>(c->prep04);
    if (!R || !R->ready) ORIP_FAIL(c, "orip_contours_prepare has not run");
    if (layer < 0 || layer >= R->K) ORIP_FAIL(c, "bad layer %d (prepared for %d layers)", layer, R->K);
    if (R->M == 0 || R->layer_first[layer] == R->layer_first[layer + 1]) return 0;
    ORIP_LANE(c, layer + 1);
    if (!R->launched[layer]) ORIP_TRY(trace_launch(c, *R, layer, 64));      // (normally enqueued by orip_contours_prepare already)
    ORIP_TRY(trace_finish(c, *R, layer));
    if (sync) HIPC(c, hipStreamSynchronize(LN(c).stream));
    return 0;
}
extern "C" int orip_contours_layer(orip_ctx* c, int layer) {
    orip_enter(c);
    return orip_contours_layer_impl(c, layer, true);
}

extern "C" int orip_find_contours(orip_ctx* c) {
    orip_enter(c);
    ORIP_TRY(orip_contours_prepare(c));
    Prep04& R = *static_cast<Prep04*>(c->prep04);
    if (R.M == 0) return 0;
    // every layer's trace is enqueued on its own stream first, so the long serial walks of all layers overlap
    for (int l = 0; l < R.K; l++) if (R.layer_first[l] != R.layer_first[l + 1] && !R.launched[l]) { ORIP_LANE(c, l + 1); ORIP_TRY(trace_launch(c, R, l, 64)); }
    for (int l = 0; l < R.K; l++) if (R.launched[l]) { ORIP_LANE(c, l + 1); ORIP_TRY(trace_finish(c, R, l)); HIPC(c, hipStreamSynchronize(LN(c).stream)); }
    return 0;
}

extern "C" int orip_get_skeleton(orip_ctx* c, int layer, uint8_t* out) {
    orip_enter(c);
    if (!c->skel.p || layer < 0 || layer >= c->K) ORIP_FAIL(c, "no skeleton for layer %d", layer);
    size_t plane = (size_t)c->H * c->W;
    HIPC(c, hipMemcpyAsync(out, c->skel.as<u8>() + plane * layer, plane, hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    return 0;
}
