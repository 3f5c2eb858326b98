// csrc/vector10.hip -- stage 10 (10_dedup_cross_basic.py main, 10:212-278) on gfx950.
// Layers are visited dark -> light against one cumulative forbidden raster (10:215,236-267):
//   cut      : every polyline is walked in ~1 px steps (10:142-177); all step points of all polylines of the layer are
//              evaluated in parallel against the raster as left by the PREVIOUS layers (the raster does not change while
//              a layer is cut), runs of >= 2 free points become the new polylines
//   jumps    : _split_on_long_jumps (10:49-63) is the identity on cut output (consecutive points are <= 1 px apart)
//   taps     : _tiny_and_taps (10:99-118) with cv::minEnclosingCircle only where the bbox cannot decide
//   reorder  : _reorder_for_travel (10:69-97) -> k_greedy_nn
//   paint    : cv2.polylines(thickness=120) over vertex chains whose consecutive vertices are <= sqrt(2) apart equals, on
//              the integer grid, the union of radius-60 discs at the vertices (DESIGN.md "stage 10 paint"), evaluated as an
//              exact two-pass disc dilation (row pass with wave ballots, column pass)
//   taps seq : sequential accept + immediate stamp (10:259-267) as one workgroup walking the tap list
#include "vec_common.h"
#include <chrono>
#include <cstdlib>
#define ORIP_PAD 64

// ---- cut ----
__global__ __launch_bounds__(256) void k_cut_counts(const int64_t* __restrict__ off, const int32_t* __restrict__ pts, int64_t n_polys, double step_px,
                                                     unsigned* __restrict__ cnt, uint8_t* __restrict__ first_of_poly) {
    // one block per polyline (grid-stride), threads over its points
    for (int64_t p = blockIdx.x; p < n_polys; p += gridDim.x) {
        int64_t b = off[p], e = off[p + 1];
        for (int64_t i = b + threadIdx.x; i < e; i += 256) {
            unsigned c = 0; uint8_t f = 0;
            if (e - b >= 2) {
                if (i == b) { c = 1; f = 1; }
                else {
                    float vx = (float)pts[2 * i] - (float)pts[2 * i - 2], vy = (float)pts[2 * i + 1] - (float)pts[2 * i - 1];
                    double L = (double)(float)sqrt((double)vx * (double)vx + (double)vy * (double)vy);
                    if (L > 1e-6) { double q = ceil(L / fmax(1.0, step_px)); c = (unsigned)fmax(1.0, q); }
                }
            }
            cnt[i] = c; first_of_poly[i] = f;
        }
    }
}

// slot s -> (point index i, step k): i = upper_bound(base, s) - 1
__device__ __forceinline__ int64_t ub_u32(const unsigned* a, int64_t n, unsigned v) {   // first index with a[idx] > v
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(256) void k_cut_slots(const int32_t* __restrict__ pts, const unsigned* __restrict__ cnt, const unsigned* __restrict__ base, int64_t n_pts,
                                                    const uint8_t* __restrict__ first_of_poly, unsigned n_slots, const u8* __restrict__ forb, int H, int W,
                                                    int2* __restrict__ spt, uint8_t* __restrict__ sflag /* bit0 free, bit1 poly start */) {
    unsigned s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_slots) return;
    int64_t i = ub_u32(base, n_pts, s) - 1;
    while (cnt[i] == 0) i--;            // base[] repeats over zero-count points; the owner is the last one with cnt > 0 and base <= s
    float qx, qy; uint8_t fl = 0;
    if (first_of_poly[i]) { qx = (float)pts[2 * i]; qy = (float)pts[2 * i + 1]; fl = 2; }
    else {
        unsigned k = s - base[i] + 1, n = cnt[i];
        float p0x = (float)pts[2 * i - 2], p0y = (float)pts[2 * i - 1];
        float vx = __fsub_rn((float)pts[2 * i], p0x), vy = __fsub_rn((float)pts[2 * i + 1], p0y);
        float t = (float)((double)k / (double)n);
        qx = __fadd_rn(p0x, __fmul_rn(vx, t)); qy = __fadd_rn(p0y, __fmul_rn(vy, t));
    }
    long long xi = vs::round_half_even((double)qx), yi = vs::round_half_even((double)qy);
    bool blocked = (yi >= 0 && yi < H && xi >= 0 && xi < W) && forb[(size_t)yi * W + xi] != 0;
    if (!blocked) fl |= 1;
    spt[s] = make_int2((int)qx, (int)qy);
    sflag[s] = fl;
}
__global__ __launch_bounds__(256) void k_run_starts(const uint8_t* __restrict__ sflag, unsigned n, unsigned* __restrict__ start) {
    unsigned s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    uint8_t f = sflag[s];
    start[s] = ((f & 1) && ((f & 2) || s == 0 || !(sflag[s - 1] & 1))) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_run_accum(const uint8_t* __restrict__ sflag, const unsigned* __restrict__ start, const unsigned* __restrict__ start_scan,
                                                    unsigned n, unsigned* __restrict__ rlen, unsigned* __restrict__ rbegin) {
    unsigned s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    if (!(sflag[s] & 1)) return;
    unsigned rid = start_scan[s] + start[s] - 1;      // inclusive scan - 1
    atomicAdd(&rlen[rid], 1u);
    if (start[s]) rbegin[rid] = s;
}
__global__ __launch_bounds__(256) void k_run_keep(const unsigned* __restrict__ rlen, unsigned n_runs, unsigned min_len, unsigned* __restrict__ keep) {
    unsigned r = blockIdx.x * 256 + threadIdx.x;
    if (r < n_runs) keep[r] = rlen[r] >= min_len ? 1u : 0u;
    if (r == n_runs) keep[r] = 0;
}
__global__ __launch_bounds__(256) void k_run_desc(const unsigned* __restrict__ rlen, const unsigned* __restrict__ rbegin, const unsigned* __restrict__ keep,
                                                   const unsigned* __restrict__ keep_scan, unsigned n_runs, GatherDesc* __restrict__ d) {
    unsigned r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_runs || !keep[r]) return;
    GatherDesc g; g.begin = rbegin[r]; g.len = rlen[r]; g.rev = 0; g.src = 0;
    d[keep_scan[r]] = g;
}

// Shared by stage 08-A and stage 10: turn per-slot flags (bit0 accepted, bit1 sequence start) + points into a DPolys of runs with >= 2 points
int orip_runs_to_polys(orip_ctx* c, const int2* spt, const uint8_t* sflag, unsigned n_slots, DPolys& dst) {
    dst.n = 0; dst.total = 0; dst.set_explicit();
    HIPC(c, dst.off.ensure(64)); HIPC(c, hipMemsetAsync(dst.off.p, 0, 8, LN(c).stream));
    if (n_slots == 0) return 0;
    HIPC(c, LN(c).vtmp[7].ensure((size_t)n_slots * 8 + 64));
    unsigned* start = LN(c).vtmp[7].as<unsigned>(); unsigned* start_scan = start + n_slots;
    hipLaunchKernelGGL(k_run_starts, dim3(cdiv(n_slots, 256)), dim3(256), 0, LN(c).stream, sflag, n_slots, start);
    ORIP_TRY(vscan_excl<unsigned>(c, start, start_scan, n_slots));
    unsigned a[2];
    HIPC(c, hipMemcpyAsync(&a[0], start_scan + (n_slots - 1), 4, hipMemcpyDeviceToHost, LN(c).stream));      // both words, one wait
    ORIP_TRY(vread(c, &a[1], start + (n_slots - 1)));
    unsigned n_runs = a[0] + a[1];
    if (n_runs == 0) return 0;
    HIPC(c, LN(c).vtmp[8].ensure((size_t)(n_runs + 1) * 16 + (size_t)n_runs * sizeof(GatherDesc) + 256));
    unsigned* rlen = LN(c).vtmp[8].as<unsigned>(); unsigned* rbegin = rlen + (n_runs + 1); unsigned* keep = rbegin + (n_runs + 1); unsigned* keep_scan = keep + (n_runs + 1);
    GatherDesc* desc = (GatherDesc*)(keep_scan + (n_runs + 1) + 2);
    HIPC(c, hipMemsetAsync(rlen, 0, (size_t)(n_runs + 1) * 4, LN(c).stream));
    hipLaunchKernelGGL(k_run_accum, dim3(cdiv(n_slots, 256)), dim3(256), 0, LN(c).stream, sflag, start, start_scan, n_slots, rlen, rbegin);
    hipLaunchKernelGGL(k_run_keep, dim3(cdiv(n_runs + 1, 256)), dim3(256), 0, LN(c).stream, rlen, n_runs, 2u, keep);
    ORIP_TRY(vscan_excl<unsigned>(c, keep, keep_scan, (size_t)n_runs + 1));
    unsigned n_keep = 0;
    ORIP_TRY(vread(c, &n_keep, keep_scan + n_runs));
    if (n_keep == 0) return 0;
    hipLaunchKernelGGL(k_run_desc, dim3(cdiv(n_runs, 256)), dim3(256), 0, LN(c).stream, rlen, rbegin, keep, keep_scan, n_runs, desc);
    HIPC(c, hipGetLastError());
    return vgather(c, desc, n_keep, reinterpret_cast<const int32_t*>(spt), dst);
}

// ---- _tiny_and_taps (10:99-118) ----
// cv::minEnclosingCircle (recalled OpenCV 4.x algorithm, see vec_serial.h) with one wavefront per polyline: the three nested
// loops of the incremental algorithm are "skip points that are inside, update on the first one that is not", so each of them is
// run as a 64-wide search for the next violating point; updates happen in exactly the sequential order.
__device__ __forceinline__ int wave_first(bool pred) { unsigned long long m = __ballot(pred); return m ? (__ffsll((long long)m) - 1) : -1; }
__device__ void mec_third_w(const int32_t* xy, int i, int j, vs::P2& c, float& radius, int lane) {
    const float EPS = 1.0e-4f;
    vs::P2 pi = vs::ipt(xy, i), pj = vs::ipt(xy, j);
    c.x = (pj.x + pi.x) / 2.0f; c.y = (pj.y + pi.y) / 2.0f;
    radius = (float)vs::nrm2(pj.x - pi.x, pj.y - pi.y) / 2.0f + EPS;
    for (int k0 = 0; k0 < j;) {
        int k = k0 + lane; bool viol = false;
        if (k < j) { vs::P2 pk = vs::ipt(xy, k); viol = !(vs::nrm2(c.x - pk.x, c.y - pk.y) < (double)radius); }
        int f = wave_first(viol);
        if (f < 0) { k0 += 64; continue; }
        int kk = k0 + f;
        vs::P2 nc{0, 0}; float nr = 0;
        vs::mec_circle3(pi, pj, vs::ipt(xy, kk), nc, nr);
        if (nr > 0) { radius = nr; c = nc; }
        k0 = kk + 1;
    }
}
__device__ void mec_second_w(const int32_t* xy, int i, vs::P2& c, float& radius, int lane) {
    const float EPS = 1.0e-4f;
    vs::P2 p0 = vs::ipt(xy, 0), pi = vs::ipt(xy, i);
    c.x = (p0.x + pi.x) / 2.0f; c.y = (p0.y + pi.y) / 2.0f;
    radius = (float)vs::nrm2(p0.x - pi.x, p0.y - pi.y) / 2.0f + EPS;
    for (int j0 = 1; j0 < i;) {
        int j = j0 + lane; bool viol = false;
        if (j < i) { vs::P2 pj = vs::ipt(xy, j); viol = !(vs::nrm2(c.x - pj.x, c.y - pj.y) < (double)radius); }
        int f = wave_first(viol);
        if (f < 0) { j0 += 64; continue; }
        int jj = j0 + f;
        vs::P2 nc{0, 0}; float nr = 0;
        mec_third_w(xy, i, jj, nc, nr, lane);
        if (nr > 0) { radius = nr; c = nc; }
        j0 = jj + 1;
    }
}
__device__ void mec_wave(const int32_t* xy, int n, float& cx, float& cy, float& r, int lane) {
    if (n <= 2) { vs::min_enclosing_circle(xy, n, cx, cy, r); return; }
    const float EPS = 1.0e-4f;
    vs::P2 p0 = vs::ipt(xy, 0), p1 = vs::ipt(xy, 1);
    vs::P2 c{(p0.x + p1.x) / 2.0f, (p0.y + p1.y) / 2.0f};
    float radius = (float)vs::nrm2(p0.x - p1.x, p0.y - p1.y) / 2.0f + EPS;
    for (int i0 = 2; i0 < n;) {
        int i = i0 + lane; bool viol = false;
        if (i < n) { vs::P2 pi = vs::ipt(xy, i); float d = (float)vs::nrm2(pi.x - c.x, pi.y - c.y); viol = !(d < radius); }
        int f = wave_first(viol);
        if (f < 0) { i0 += 64; continue; }
        int ii = i0 + f;
        vs::P2 nc{0, 0}; float nr = 0;
        mec_second_w(xy, ii, nc, nr, lane);
        if (nr > 0) { radius = nr; c = nc; }
        i0 = ii + 1;
    }
    cx = c.x; cy = c.y; r = radius;
}
// one wavefront per polyline
__global__ __launch_bounds__(64) void k_tiny_taps10(const int64_t* __restrict__ off, const int32_t* __restrict__ pts, int64_t n_polys, orip_params10 P, const PolyFeat* __restrict__ feat,
                                                     unsigned* __restrict__ is_tap, unsigned* __restrict__ is_keep, int2* __restrict__ tap_xy) {
    const int lane = threadIdx.x;
    for (int64_t i = blockIdx.x; i <= n_polys; i += gridDim.x) {
        if (i == n_polys) { if (lane == 0) { is_tap[i] = 0; is_keep[i] = 0; } continue; }
        const int32_t* p = pts + 2 * off[i]; int64_t n = off[i + 1] - off[i];
        const int32_t x0 = feat[i].x0, x1 = feat[i].x1, y0 = feat[i].y0, y1 = feat[i].y1;
        unsigned tap = 0, keep = 0; int2 txy = make_int2(0, 0);
        double big = fmax(P.tap_diam, P.min_keep) + 2.0;
        const double ext = (double)max(x1 - x0, y1 - y0);
        if (ext > big) keep = 1;                                 // enclosing diameter >= bbox extent > both thresholds: no circle needed
        else if (n > (int64_t)P.tap_max_v && ext >= P.min_keep + 2.0) keep = 1;     // too many vertices for a tap (10:108-112) and certainly not tiny: no circle either
        else {
            float cx, cy, r; mec_wave(p, (int)n, cx, cy, r, lane);
            double d = 2.0 * (double)r;
            if (d <= P.tap_diam && n <= (int64_t)P.tap_max_v) {      // the vertex test is evaluated after the perimeter in the reference but decides alone
                double per = vs::arc_length(p, n, false);
                if (per <= P.tap_max_per) { tap = 1; txy = make_int2((int)vs::round_half_even((double)cx), (int)vs::round_half_even((double)cy)); }
            }
            if (!tap && d >= P.min_keep) keep = 1;
        }
        if (lane == 0) { is_tap[i] = tap; is_keep[i] = keep; if (tap) tap_xy[i] = txy; }
    }
}
__global__ __launch_bounds__(256) void k_compact_sel(const unsigned* __restrict__ flag, const unsigned* __restrict__ scan, int64_t n, const int64_t* __restrict__ off,
                                                      GatherDesc* __restrict__ d, const int2* __restrict__ tap_xy, int2* __restrict__ taps_out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !flag[i]) return;
    if (d) { GatherDesc g; g.begin = off[i]; g.len = off[i + 1] - off[i]; g.rev = 0; g.src = (int32_t)i; d[scan[i]] = g; }
    if (taps_out) taps_out[scan[i]] = tap_xy[i];
}

// ---- paint: exact disc dilation of the vertex set ----
// occ: one byte per 64x32 block of the padded raster, set when the block holds a seed (lets the row / column passes skip empty regions)
__global__ __launch_bounds__(256) void k_seed_mark(const int2* __restrict__ pts, int64_t n, u8* __restrict__ seeds, int Hp, int Wp, u8* __restrict__ occ, int occ_w) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int x = pts[i].x + ORIP_PAD, y = pts[i].y + ORIP_PAD;
    if (x >= 0 && x < Wp && y >= 0 && y < Hp) { seeds[(size_t)y * Wp + x] = 1; occ[(size_t)(y >> 5) * occ_w + (x >> 6)] = 1; }
}
// one wave per padded row: horizontal distance to the nearest seed of the row, capped at 255
__global__ __launch_bounds__(64) void k_row_hdist(const u8* __restrict__ seeds, u8* __restrict__ hd, int Hp, int Wp) {
    int y = blockIdx.x; const int lane = threadIdx.x;
    const u8* s = seeds + (size_t)y * Wp; u8* o = hd + (size_t)y * Wp;
    int nch = (Wp + 63) >> 6;
    int last = -100000;
    for (int c = 0; c < nch; c++) {
        int x = (c << 6) + lane;
        bool sd = x < Wp && s[x];
        unsigned long long b = __ballot(sd);
        unsigned long long m = b & ((lane == 63) ? ~0ULL : ((2ULL << lane) - 1ULL));
        int pos = m ? ((c << 6) + 63 - __clzll((long long)m)) : last;
        int d = x - pos;
        if (x < Wp) o[x] = (u8)min(d, 255);
        if (b) last = (c << 6) + 63 - __clzll((long long)b);
    }
    int nxt = 100000000;
    for (int c = nch - 1; c >= 0; c--) {
        int x = (c << 6) + lane;
        bool sd = x < Wp && s[x];
        unsigned long long b = __ballot(sd);
        unsigned long long m = b & (~0ULL << lane);
        int pos = m ? ((c << 6) + __ffsll((long long)m) - 1) : nxt;
        int d = pos - x;
        if (x < Wp) { int cur = o[x]; o[x] = (u8)min(cur, min(d, 255)); }
        if (b) nxt = (c << 6) + __ffsll((long long)b) - 1;
    }
}
// column pass: pixel (x,y) is covered iff some row y+dy (|dy| <= r) has a seed within sqrt(r^2 - dy^2) of column x.  The hd rows a 64x32 tile
// needs (r above and below) are staged in LDS once; every output then reads its column of the staged tile.
#define CC_TX 64
#define CC_TY 32
#define CC_RMAX 62
__global__ __launch_bounds__(256) void k_col_cover(const u8* __restrict__ hd, u8* __restrict__ forb, int H, int W, int Hp, int Wp, int r, const u8* __restrict__ occ, int occ_w, int occ_h) {
    __shared__ u8 T[CC_TY + 2 * CC_RMAX][CC_TX];
    const int x0 = blockIdx.x * CC_TX, y0 = blockIdx.y * CC_TY;
    {   // any seed within reach of this tile?  padded block coordinates of the tile: (x0+PAD)/64 = bx+1, rows (y0+PAD-r .. y0+PAD+TY+r)
        int obx = (x0 + ORIP_PAD) >> 6, oby0 = max(0, (y0 + ORIP_PAD - r) >> 5), oby1 = min(occ_h - 1, (y0 + ORIP_PAD + CC_TY - 1 + r) >> 5);
        bool any = false;
        for (int by = oby0; by <= oby1 && !any; by++) for (int bx = max(0, obx - 1); bx <= min(occ_w - 1, obx + 1); bx++) if (occ[(size_t)by * occ_w + bx]) { any = true; break; }
        if (!any) return;
    }
    const int rows = CC_TY + 2 * r;
    for (int i = threadIdx.x; i < rows * (CC_TX / 4); i += 256) {
        int ty = i / (CC_TX / 4), tx4 = (i % (CC_TX / 4)) * 4;
        int yy = y0 - r + ty + ORIP_PAD;                       // padded row
        uint32_t v = 0xffffffffu;
        if (yy >= 0 && yy < Hp) {
            int xx = x0 + tx4 + ORIP_PAD;
            if (xx + 3 < Wp && ((Wp & 3) == 0)) v = *reinterpret_cast<const uint32_t*>(hd + (size_t)yy * Wp + xx);
            else { v = 0; for (int j = 0; j < 4; j++) { int xj = xx + j; uint32_t b = (xj < Wp) ? hd[(size_t)yy * Wp + xj] : 255u; v |= b << (8 * j); } }
        }
        *reinterpret_cast<uint32_t*>(&T[ty][tx4]) = v;
    }
    __syncthreads();
    const int r2 = r * r;
    for (int i = threadIdx.x; i < CC_TX * CC_TY; i += 256) {
        int tx = i % CC_TX, ty = i / CC_TX;
        int x = x0 + tx, y = y0 + ty;
        if (x >= W || y >= H) continue;
        size_t o = (size_t)y * W + x;
        if (forb[o]) continue;
        bool cov = false;
        for (int dy = -r; dy <= r; dy++) { int h = T[ty + r + dy][tx]; if (h * h + dy * dy <= r2) { cov = true; break; } }
        if (cov) forb[o] = 255;
    }
}
// Discs of radius r around the vertices of a chain, written as set differences: vertex i only writes the part of its disc that the disc of vertex
// i - 1 does not hold (for the unit steps of the stage's lines a sickle of a few pixels per row; the full disc when the two do not overlap).  By
// induction the union of what is written is the union of the discs, whatever the order the threads run in.  One thread per (vertex, row).
__global__ __launch_bounds__(256) void k_stamp_chain(const int2* __restrict__ pts, int64_t n, int r, u8* __restrict__ forb, int H, int W) {
    __shared__ int hw[2 * CC_RMAX + 4];                        // half width of the disc at row offset d: floor(sqrt(r^2 - d^2))
    const int side = 2 * r + 1;
    for (int d = threadIdx.x; d < side; d += 256) {
        const int dy = d - r; const int q = r * r - dy * dy;
        int w = (int)sqrtf((float)q);
        while (w * w > q) w--;
        while ((w + 1) * (w + 1) <= q) w++;
        hw[d] = w;
    }
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * side) return;
    const int64_t i = t / side; const int d = (int)(t - i * side);
    const int2 v = pts[i];
    const int y = v.y + d - r;
    if (y < 0 || y >= H) return;
    int lo = v.x - hw[d], hi = v.x + hw[d];
    int plo = 1, phi = 0;                                      // span of the previous vertex's disc on this row (empty)
    if (i > 0) {
        const int2 p = pts[i - 1];
        const int dp = y - p.y;
        if (dp >= -r && dp <= r) { plo = p.x - hw[dp + r]; phi = p.x + hw[dp + r]; }
    }
    lo = max(lo, 0); hi = min(hi, W - 1);
    u8* row = forb + (size_t)y * W;
    if (plo > phi) { for (int x = lo; x <= hi; x++) row[x] = 255; return; }
    for (int x = lo; x <= min(hi, plo - 1); x++) row[x] = 255;
    for (int x = max(lo, phi + 1); x <= hi; x++) row[x] = 255;
}
__global__ __launch_bounds__(256) void k_stamp_discs(const int2* __restrict__ taps, int n, int r, u8* __restrict__ forb, int H, int W) {
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        int cx = taps[t].x, cy = taps[t].y; int side = 2 * r + 1;
        for (int i = threadIdx.x; i < side * side; i += 256) {
            int dx = i % side - r, dy = i / side - r;
            int x = cx + dx, y = cy + dy;
            if (x < 0 || x >= W || y < 0 || y >= H) continue;
            if (dx * dx + dy * dy <= r * r) forb[(size_t)y * W + x] = 255;
        }
    }
}
// sequential taps (10:259-267): accept iff the centre is free of the raster AND of every disc accepted before it in this layer
__global__ __launch_bounds__(1024) void k_taps_sequential(const int2* __restrict__ seq, int n, int r, const u8* __restrict__ forb, int H, int W,
                                                           int2* __restrict__ acc, int* __restrict__ n_acc_out) {
    __shared__ int nacc, hit;
    if (threadIdx.x == 0) nacc = 0;
    __syncthreads();
    const long long r2 = (long long)r * r;
    for (int i = 0; i < n; i++) {
        int x = seq[i].x, y = seq[i].y;
        bool inside = (y >= 0 && y < H && x >= 0 && x < W);
        if (threadIdx.x == 0) hit = (inside && forb[(size_t)y * W + x] != 0) ? 1 : 0;
        __syncthreads();
        if (inside && !hit) {
            int na = nacc;
            for (int j = threadIdx.x; j < na; j += 1024) {
                long long dx = acc[j].x - x, dy = acc[j].y - y;
                if (dx * dx + dy * dy <= r2) hit = 1;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0 && !hit) { acc[nacc] = make_int2(x, y); nacc++; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_acc_out = nacc;
}

// the same sequential filter by one wavefront: the raster test of every candidate is order-free (the raster only changes after the
// loop) and is done up front; the candidates and the accepted list then live in LDS and each candidate is checked against the
// accepted ones 64 at a time
__global__ __launch_bounds__(64) void k_taps_wave(const int2* __restrict__ seq, int n, int r, const u8* __restrict__ forb, int H, int W,
                                                   int2* __restrict__ acc_out, int* __restrict__ n_acc_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    int2* cand = reinterpret_cast<int2*>(smem); int2* acc = cand + n;
    uint8_t* st = reinterpret_cast<uint8_t*>(acc + n);            // 0: rejected by the raster, 1: inside, 2: outside the canvas (accepted as is)
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += 64) {
        const int2 p = seq[i]; cand[i] = p;
        const bool inside = (p.y >= 0 && p.y < H && p.x >= 0 && p.x < W);
        st[i] = inside ? (forb[(size_t)p.y * W + p.x] != 0 ? 0 : 1) : 2;
    }
    __syncthreads();
    const long long r2 = (long long)r * r;
    int nacc = 0;
    for (int i = 0; i < n; i++) {
        const uint8_t s = st[i];
        if (s == 0) continue;
        const int2 p = cand[i];
        bool hit = false;
        if (s == 1) {
            for (int j0 = 0; j0 < nacc && !hit; j0 += 64) {
                const int j = j0 + lane; bool h = false;
                if (j < nacc) { const int2 a = acc[j]; const long long dx = a.x - p.x, dy = a.y - p.y; h = dx * dx + dy * dy <= r2; }
                hit = __ballot(h) != 0;
            }
        }
        if (!hit) { if (lane == 0) acc[nacc] = p; nacc++; }
    }
    for (int i = lane; i < nacc; i += 64) acc_out[i] = acc[i];
    if (lane == 0) *n_acc_out = nacc;
}

// Stage 10 keeps one cumulative "forbidden" raster while it walks the layers from dark to light (10:166-212).  begin clears it,
// every orip_dedup_cross_layer call handles the next layer of that order; the calls use their own lane (stream + scratch), so the
// early layers can be processed while later ones are still in stages 04-08 on their lanes.
extern "C" int orip_dedup_cross_begin(orip_ctx* c, const orip_params10* prm) {
    orip_enter(c);
    if (!prm) ORIP_FAIL(c, "bad arguments");
    const orip_params10 P = *prm;
    const int W = P.W, H = P.H;
    if (W <= 0 || H <= 0 || W > 16383 || H > 16383) ORIP_FAIL(c, "canvas %dx%d out of range", W, H);
    if (P.max_jump < 4.0) ORIP_FAIL(c, "max_join_jump_px < 4 is not supported (cut output is assumed jump-free)");
    const int rad_lines = (int)std::max<long long>(1, vs::round_half_even(P.D_lines)) / 2;
    const int rad_taps = (int)std::max<long long>(1, vs::round_half_even(P.D_taps / 2.0));
    if (rad_lines > ORIP_PAD - 2 || rad_taps > 200) ORIP_FAIL(c, "brush radius %d/%d too large for the padded raster", rad_lines, rad_taps);
    ORIP_LANE(c, ORIP_LANE_CROSS);
    HIPC(c, LN(c).canvas.ensure((size_t)W * H + 64));
    HIPC(c, hipMemsetAsync(LN(c).canvas.p, 0, (size_t)W * H, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    c->p10 = P; c->p10_ready = true;
    return 0;
}

// input lists from layer slot `src_layer` (LINES/TAPS_INTRA), output and reporting under `layer`: lets a process that holds its own
// layers under local indices feed stage 10 in global layer order (multi-GPU layer sharding)
static int dedup_cross_layer_impl(orip_ctx* c, int src_layer, int layer, bool reorder_now);
extern "C" int orip_dedup_cross_layer_from(orip_ctx* c, int src_layer, int layer) {
    orip_enter(c); return dedup_cross_layer_impl(c, src_layer, layer, true); }
// The same without the travel reorder of the kept lines (10:253, _reorder_for_travel): nothing later in stage 10 depends on their order
// (the paint is a union, the taps keep their own order), so the one-wave greedy chain need not sit on the serial layer-after-layer path of
// this stage.  LINES_CROSS of `layer` is left in cut order and marked; orip_plot_order(layer) reorders it first, on the layer's own lane.
extern "C" int orip_dedup_cross_layer_deferred(orip_ctx* c, int src_layer, int layer) {
    orip_enter(c); return dedup_cross_layer_impl(c, src_layer, layer, false); }
static int dedup_cross_layer_impl(orip_ctx* c, int src_layer, int layer, bool reorder_now) {
    if (src_layer < 0 || src_layer >= ORIP_MAX_LAYERS) ORIP_FAIL(c, "bad layer %d", src_layer);
    if (!c->p10_ready) ORIP_FAIL(c, "orip_dedup_cross_begin has not run");
    const orip_params10 P = c->p10;
    const int W = P.W, H = P.H;
    ORIP_LANE(c, ORIP_LANE_CROSS);
    const int Wp = W + 2 * ORIP_PAD, Hp = H + 2 * ORIP_PAD;
    const int rad_lines = (int)std::max<long long>(1, vs::round_half_even(P.D_lines)) / 2;
    const int rad_taps = (int)std::max<long long>(1, vs::round_half_even(P.D_taps / 2.0));
    u8* forb = LN(c).canvas.as<u8>();
    const int occ_w = (Wp + 63) >> 6, occ_h = (Hp + 31) >> 5;
    HIPC(c, LN(c).vtmp[9].ensure((size_t)Wp * Hp * 2 + (size_t)occ_w * occ_h + 64));
    u8* seeds = LN(c).vtmp[9].as<u8>(); u8* hd = seeds + (size_t)Wp * Hp; u8* occ = hd + (size_t)Wp * Hp;
    const bool tdbg = getenv("ORIP_TIME10") != nullptr;
    auto now = [&]() { hipStreamSynchronize(LN(c).stream); return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    {

        auto t0 = tdbg ? now() : std::chrono::steady_clock::time_point();
        if (layer < 0 || layer >= ORIP_MAX_LAYERS) ORIP_FAIL(c, "bad layer %d", layer);
        DPolys& Lin = c->polys[ORIP_SLOT_LINES_INTRA][src_layer]; DPolys& Lout = c->polys[ORIP_SLOT_LINES_CROSS][layer];
        DTaps& Tin = c->taps[ORIP_TAPS_INTRA][src_layer]; DTaps& Tout = c->taps[ORIP_TAPS_CROSS][layer];
        // ---- 1) cut
        DPolys& cut = LN(c).tp[4]; DPolys& keepl = LN(c).tp[5];   // persistent temporaries of this lane
        cut.n = 0; cut.total = 0; cut.set_explicit();
        ORIP_TRY(orip_polys_materialize(c, Lin));      // stage 08 leaves explicit lists; a walk-coded one set up by hand is expanded first
        if (Lin.n > 0 && Lin.total > 0) {
            if (Lin.total > 0x7fffffff) ORIP_FAIL(c, "layer too large");
            HIPC(c, LN(c).vtmp[0].ensure((size_t)(Lin.total + 1) * 9 + 64));
            unsigned* cnt = LN(c).vtmp[0].as<unsigned>(); unsigned* base = cnt + (Lin.total + 1); uint8_t* fop = (uint8_t*)(base + (Lin.total + 1));
            HIPC(c, hipMemsetAsync(cnt + Lin.total, 0, 4, LN(c).stream));
            hipLaunchKernelGGL(k_cut_counts, dim3((unsigned)std::min<int64_t>(Lin.n, 65535)), dim3(256), 0, LN(c).stream, Lin.off.as<int64_t>(), Lin.pts.as<int32_t>(), Lin.n, P.step_px, cnt, fop);
            ORIP_TRY(vscan_excl<unsigned>(c, cnt, base, (size_t)Lin.total + 1));
            unsigned n_slots = 0;
            ORIP_TRY(vread(c, &n_slots, base + Lin.total));
            if (n_slots) {
                HIPC(c, LN(c).vtmp[1].ensure((size_t)n_slots * 9 + 64));
                int2* spt = LN(c).vtmp[1].as<int2>(); uint8_t* sflag = (uint8_t*)(spt + n_slots);
                { ProfScope ps(c, "k_cut_slots"); hipLaunchKernelGGL(k_cut_slots, dim3(cdiv(n_slots, 256)), dim3(256), 0, LN(c).stream, Lin.pts.as<int32_t>(), cnt, base, Lin.total, fop, n_slots, forb, H, W, spt, sflag); }
                ORIP_TRY(orip_runs_to_polys(c, spt, sflag, n_slots, cut));
            }
        }
        auto t1 = tdbg ? now() : t0;
        // ---- 2,3) jumps are the identity; tiny lines -> taps / dropped
        int64_t n_tap_lines = 0;
        keepl.n = 0; keepl.total = 0; keepl.set_explicit();
        HIPC(c, LN(c).vtmp[2].ensure((size_t)(cut.n + 1) * (16 + 8 + sizeof(GatherDesc)) + 256));
        unsigned* is_tap = LN(c).vtmp[2].as<unsigned>(); unsigned* is_keep = is_tap + (cut.n + 1); unsigned* tap_scan = is_keep + (cut.n + 1); unsigned* keep_scan = tap_scan + (cut.n + 1);
        int2* tap_xy = (int2*)(keep_scan + (cut.n + 1)); GatherDesc* kd = (GatherDesc*)(tap_xy + (cut.n + 1));
        int64_t n_seq = Tin.n;
        if (cut.n > 0) {
            HIPC(c, LN(c).vtmp[10].ensure((size_t)cut.n * sizeof(PolyFeat) + 64));
            PolyFeat* cfeat = LN(c).vtmp[10].as<PolyFeat>();
            ORIP_TRY(vfeatures(c, cut, 0, cfeat));
            hipLaunchKernelGGL(k_tiny_taps10, dim3((unsigned)std::min<int64_t>(cut.n + 1, 65535)), dim3(64), 0, LN(c).stream, cut.off.as<int64_t>(), cut.pts.as<int32_t>(), cut.n, P, cfeat, is_tap, is_keep, tap_xy);
            ORIP_TRY(vscan_excl<unsigned>(c, is_tap, tap_scan, (size_t)cut.n + 1));
            ORIP_TRY(vscan_excl<unsigned>(c, is_keep, keep_scan, (size_t)cut.n + 1));
            unsigned a = 0, b = 0;
            HIPC(c, hipMemcpyAsync(&a, tap_scan + cut.n, 4, hipMemcpyDeviceToHost, LN(c).stream));      // both counts, one wait
            ORIP_TRY(vread(c, &b, keep_scan + cut.n));
            n_tap_lines = a;
            if (b) {
                hipLaunchKernelGGL(k_compact_sel, dim3(cdiv(cut.n, 256)), dim3(256), 0, LN(c).stream, is_keep, keep_scan, cut.n, cut.off.as<int64_t>(), kd, (const int2*)nullptr, (int2*)nullptr);
                ORIP_TRY(vgather(c, kd, b, cut.pts.as<int32_t>(), keepl));
            }
        }
        // taps_seq = taps_in + taps_from_lines
        n_seq = Tin.n + n_tap_lines;
        HIPC(c, LN(c).vtmp[3].ensure((size_t)(n_seq + 1) * 16 + 64));
        int2* seq = LN(c).vtmp[3].as<int2>(); int2* acc = seq + (n_seq + 1);
        if (Tin.n) HIPC(c, hipMemcpyAsync(seq, Tin.xy.p, (size_t)Tin.n * 8, hipMemcpyDeviceToDevice, LN(c).stream));
        if (n_tap_lines) hipLaunchKernelGGL(k_compact_sel, dim3(cdiv(cut.n, 256)), dim3(256), 0, LN(c).stream, is_tap, tap_scan, cut.n, cut.off.as<int64_t>(), (GatherDesc*)nullptr, tap_xy, seq + Tin.n);
        auto t2 = tdbg ? now() : t0;
        // ---- 4) reorder (or hand the kept lines over as they are: orip_dedup_cross_layer_deferred)
        if (reorder_now) { ORIP_TRY(vreorder(c, keepl, Lout, 10)); c->cross_unordered[layer] = false; }
        else {
            if (keepl.n == 0) { HIPC(c, keepl.off.ensure(64)); HIPC(c, hipMemsetAsync(keepl.off.p, 0, 8, LN(c).stream)); }
            // The buffers change hands (lane temporary <-> the layer's slot).  Whatever comes back is grown to the largest list seen so far NOW,
            // while nothing is running on it, so that no later layer finds a too small buffer in the middle of its serial tail: growing means
            // hipFree, and hipFree waits for every stream of the device (4 of them cost 4.5 ms per step before this).
            c->hw_cross_off = std::max(c->hw_cross_off, (size_t)(keepl.n + 1) * 8 + 64); c->hw_cross_pts = std::max(c->hw_cross_pts, (size_t)std::max<int64_t>(keepl.total, 1) * 8 + 64);
            std::swap(keepl.off, Lout.off); std::swap(keepl.pts, Lout.pts); Lout.n = keepl.n; Lout.total = keepl.total; Lout.set_explicit();
            HIPC(c, keepl.off.ensure(c->hw_cross_off)); HIPC(c, keepl.pts.ensure(c->hw_cross_pts));
            c->cross_unordered[layer] = true;
        }
        auto t3 = tdbg ? now() : t0;
        // ---- 5) paint lines (exact disc dilation of all vertices)
        // The usual case after stage 08 (tens of thousands of vertices on a 100-Mpixel canvas): discs written directly, each vertex only what its
        // predecessor's disc does not hold (k_stamp_chain).  The separable passes below cost three sweeps of the whole canvas whatever the number
        // of vertices and only win for vertex counts in the order of the canvas.  Same predicate either way: (x - vx)^2 + (y - vy)^2 <= r^2.
        const long long chain_threads = (long long)Lout.total * (2 * rad_lines + 1);
        if (Lout.total > 0 && rad_lines <= CC_RMAX && chain_threads <= 4ll * W * H && chain_threads < (1ll << 39) && !getenv("ORIP_PAINT_SEPARABLE")) {
            ProfScope ps(c, "k_stamp_chain");
            hipLaunchKernelGGL(k_stamp_chain, dim3((unsigned)((chain_threads + 255) / 256)), dim3(256), 0, LN(c).stream, reinterpret_cast<const int2*>(Lout.pts.p), Lout.total, rad_lines, forb, H, W);
        } else if (Lout.total > 0) {
            HIPC(c, hipMemsetAsync(seeds, 0, (size_t)Wp * Hp, LN(c).stream));
            HIPC(c, hipMemsetAsync(occ, 0, (size_t)occ_w * occ_h, LN(c).stream));
            hipLaunchKernelGGL(k_seed_mark, dim3(cdiv(Lout.total, 256)), dim3(256), 0, LN(c).stream, reinterpret_cast<const int2*>(Lout.pts.p), Lout.total, seeds, Hp, Wp, occ, occ_w);
            { ProfScope ps(c, "k_row_hdist"); hipLaunchKernelGGL(k_row_hdist, dim3(Hp), dim3(64), 0, LN(c).stream, seeds, hd, Hp, Wp); }
            { ProfScope ps(c, "k_col_cover"); hipLaunchKernelGGL(k_col_cover, dim3(cdiv(W, CC_TX), cdiv(H, CC_TY)), dim3(256), 0, LN(c).stream, hd, forb, H, W, Hp, Wp, rad_lines, occ, occ_w, occ_h); }
        }
        auto t4 = tdbg ? now() : t0;
        // ---- 6) sequential taps
        Tout.n = 0;
        HIPC(c, Tout.xy.ensure((size_t)std::max<int64_t>(n_seq, 1) * 8 + 64));
        if (n_seq > 0) {
            int* d_n = LN(c).flags.as<int>() + 44;
            const size_t lds_t = (size_t)n_seq * 17 + 64;
            if (lds_t <= 150 * 1024 && !getenv("ORIP_TAPS_1WG")) {
                static std::once_flag attr_once;            // several layer threads may arrive here together
                static std::atomic<int> attr_err{0};
                std::call_once(attr_once, [&] { orip_max_lds(k_taps_wave, 150 * 1024, attr_err); });
                if (attr_err.load()) ORIP_FAIL(c, "hipFuncSetAttribute(k_taps_wave) failed: %s", hipGetErrorString((hipError_t)attr_err.load()));
                ProfScope ps(c, "k_taps_sequential");
                hipLaunchKernelGGL(k_taps_wave, dim3(1), dim3(64), lds_t, LN(c).stream, seq, (int)n_seq, rad_taps, forb, H, W, acc, d_n);
            } else { ProfScope ps(c, "k_taps_sequential"); hipLaunchKernelGGL(k_taps_sequential, dim3(1), dim3(1024), 0, LN(c).stream, seq, (int)n_seq, rad_taps, forb, H, W, acc, d_n); }
            int na = 0;
            ORIP_TRY(vread(c, &na, d_n));
            Tout.n = na;
            if (na) {
                HIPC(c, hipMemcpyAsync(Tout.xy.p, acc, (size_t)na * 8, hipMemcpyDeviceToDevice, LN(c).stream));
                hipLaunchKernelGGL(k_stamp_discs, dim3((unsigned)std::min(na, 4096)), dim3(256), 0, LN(c).stream, acc, na, rad_taps, forb, H, W);
            }
        }
        HIPC(c, hipGetLastError());
        HIPC(c, hipStreamSynchronize(LN(c).stream));
        if (tdbg) { auto t5 = now(); fprintf(stderr, "[time10] layer %d: cut %.2f  tiny/taps %.2f  reorder %.2f  paint %.2f  taps %.2f ms (lines in %lld pts %lld -> out %lld pts %lld)\n", layer, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5), (long long)Lin.n, (long long)Lin.total, (long long)Lout.n, (long long)Lout.total); }
    }
    return 0;
}

extern "C" int orip_dedup_cross_layer(orip_ctx* c, int layer) {
    orip_enter(c); return orip_dedup_cross_layer_from(c, layer, layer); }

extern "C" int orip_dedup_cross(orip_ctx* c, const int32_t* order, int n_layers, const orip_params10* prm) {
    orip_enter(c);
    if (!prm || n_layers < 0 || n_layers > ORIP_MAX_LAYERS) ORIP_FAIL(c, "bad arguments");
    ORIP_TRY(orip_dedup_cross_begin(c, prm));
    for (int li = 0; li < n_layers; li++) ORIP_TRY(orip_dedup_cross_layer(c, order[li]));
    return 0;
}

