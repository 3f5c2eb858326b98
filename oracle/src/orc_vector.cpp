// oracle/src/orc_vector.cpp -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// CPU restatement of the vector half of the hot path: stage 05 (scale), 07 (sort), 08 (intra-layer
// dedup), 10 (cross-layer dedup), 12 (plot order).  dtype / rounding rules follow SURVEY App. A.5.
// OpenCV drawing / shape primitives are restated from SURVEY App. B.8-B.9 (unverified vs real OpenCV).
#include "orc_common.h"
#include <cstring>
#include <cfloat>
#include <deque>
#include <map>
#include <unordered_map>

namespace orc {

// ----------------------------------------------------------------------------------------------
// Drawing primitives.  cv2.line / cv2.polylines with thickness t (LINE_8) are restated as the
// union of capsules of radius t/2 around each segment, evaluated exactly in
// integer arithmetic:  pixel p is painted  <=>  dist(p, segment ab)^2 <= r^2.
// cv2.circle(filled, r) is restated as the disc |p-c|^2 <= r^2.
// ----------------------------------------------------------------------------------------------
static inline bool in_capsule(int64_t px, int64_t py, int64_t ax, int64_t ay, int64_t bx, int64_t by, int64_t r2) {
    int64_t dx = bx - ax, dy = by - ay, qx = px - ax, qy = py - ay;
    int64_t L2 = dx * dx + dy * dy, q2 = qx * qx + qy * qy;
    if (L2 == 0) return q2 <= r2;
    int64_t t = qx * dx + qy * dy;
    if (t <= 0) return q2 <= r2;
    if (t >= L2) { int64_t ex = px - bx, ey = py - by; return ex * ex + ey * ey <= r2; }
    // q2 - t^2/L2 <= r2   <=>   q2*L2 - t*t <= r2*L2   (all < 2^63 for canvas coordinates)
    return q2 * L2 - t * t <= r2 * L2;
}

void stamp_capsule(u8* mask, int H, int W, int x0, int y0, int x1, int y1, int r, u8 val) {
    int bx0 = std::max(0, std::min(x0, x1) - r), bx1 = std::min(W - 1, std::max(x0, x1) + r);
    int by0 = std::max(0, std::min(y0, y1) - r), by1 = std::min(H - 1, std::max(y0, y1) + r);
    if (bx0 > bx1 || by0 > by1) return;
    int64_t r2 = (int64_t)r * r;
    // The capsule is convex, so its intersection with a pixel row is one interval.  It is found with the SAME exact
    // integer predicate as the plain double loop (which this replaces for speed only): an inside pixel is searched next
    // to the point of the segment closest to the row, then both interval ends are bisected.
    for (int y = by0; y <= by1; y++) {
        double xc;
        if ((y0 <= y && y <= y1) || (y1 <= y && y <= y0)) xc = (y1 == y0) ? 0.5 * (x0 + x1) : x0 + (double)(x1 - x0) * (double)(y - y0) / (double)(y1 - y0);
        else xc = (std::abs(y - y0) < std::abs(y - y1)) ? x0 : x1;
        int seed = -1;
        if (y1 == y0 && y == y0) { int lo = std::max(bx0, std::min(x0, x1)), hi = std::min(bx1, std::max(x0, x1)); if (lo <= hi) seed = lo; }
        if (seed < 0) {
            int f = (int)std::floor(xc);
            for (int cnd = f - 1; cnd <= f + 2; cnd++) { if (cnd < bx0 || cnd > bx1) continue; if (in_capsule(cnd, y, x0, y0, x1, y1, r2)) { seed = cnd; break; } }
        }
        if (seed < 0) {  // the closest point may lie outside the clipped box: fall back to the exact scan of this row
            for (int x = bx0; x <= bx1; x++) if (in_capsule(x, y, x0, y0, x1, y1, r2)) { seed = x; break; }
            if (seed < 0) continue;
        }
        int lo = bx0, hi = seed;       // smallest inside x in [bx0, seed]
        while (lo < hi) { int mid = (lo + hi) >> 1; if (in_capsule(mid, y, x0, y0, x1, y1, r2)) hi = mid; else lo = mid + 1; }
        int xl = lo; lo = seed; hi = bx1;   // largest inside x in [seed, bx1]
        while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (in_capsule(mid, y, x0, y0, x1, y1, r2)) lo = mid; else hi = mid - 1; }
        memset(mask + (size_t)y * W + xl, val, (size_t)(lo - xl + 1));
    }
}

void stamp_disc(u8* mask, int H, int W, int cx, int cy, int r, u8 val) {
    stamp_capsule(mask, H, W, cx, cy, cx, cy, r, val);
}

static void stamp_polyline(u8* mask, int H, int W, const int32_t* xy, size_t n, int ox, int oy, int r) {
    for (size_t i = 0; i + 1 < n; i++)
        stamp_capsule(mask, H, W, xy[2 * i] - ox, xy[2 * i + 1] - oy, xy[2 * i + 2] - ox, xy[2 * i + 3] - oy, r, 255);
}

// cv::arcLength (07:50, 10:43).  SURVEY App. B.9: float per-edge sqrt, double accumulation.
double arc_length_i32(const int32_t* xy, size_t n, bool closed) {
    if (n <= 1) return 0.0;
    size_t last = closed ? n - 1 : 0;
    float pvx = (float)xy[2 * last], pvy = (float)xy[2 * last + 1];
    double per = 0;
    for (size_t i = 0; i < n; i++) {
        float x = (float)xy[2 * i], y = (float)xy[2 * i + 1];
        float dx = x - pvx, dy = y - pvy;
        per += std::sqrt(dx * dx + dy * dy);
        pvx = x; pvy = y;
    }
    return per;
}

// _poly_perimeter (08:25-28): float32 norms, numpy pairwise float32 sum.
float poly_perimeter_f32(const int32_t* xy, size_t n) {
    if (n < 2) return 0.f;
    std::vector<float> seg(n - 1);
    for (size_t i = 0; i + 1 < n; i++) {
        float dx = (float)xy[2 * i + 2] - (float)xy[2 * i], dy = (float)xy[2 * i + 3] - (float)xy[2 * i + 1];
        seg[i] = std::sqrt(dx * dx + dy * dy);
    }
    return np_pairwise_sum_f32(seg.data(), seg.size());
}

// 12:71-76 _poly_len: float32 diffs, float32 hypot, numpy pairwise float32 sum.
static float poly_len12_f32(const int32_t* xy, size_t n) {
    if (n < 2) return 0.f;
    std::vector<float> seg(n - 1);
    for (size_t i = 0; i + 1 < n; i++) {
        float dx = (float)xy[2 * i + 2] - (float)xy[2 * i], dy = (float)xy[2 * i + 3] - (float)xy[2 * i + 1];
        seg[i] = hypot_f32(dx, dy);
    }
    return np_pairwise_sum_f32(seg.data(), seg.size());
}

// cv::minEnclosingCircle on float points (08:212, 10:46,113).  Recalled from OpenCV 4.x shapedescr.cpp.
namespace mec {
const float EPS = 1.0e-4f;
struct P2 { float x, y; };
static inline double nrm(float dx, float dy) { return std::sqrt((double)dx * dx + (double)dy * dy); }
static void circle3(const P2* pts, P2& c, float& radius) {
    P2 v1{pts[1].x - pts[0].x, pts[1].y - pts[0].y}, v2{pts[2].x - pts[0].x, pts[2].y - pts[0].y};
    P2 m1{(pts[0].x + pts[1].x) / 2.0f, (pts[0].y + pts[1].y) / 2.0f};
    float c1 = m1.x * v1.x + m1.y * v1.y;
    P2 m2{(pts[0].x + pts[2].x) / 2.0f, (pts[0].y + pts[2].y) / 2.0f};
    float c2 = m2.x * v2.x + m2.y * v2.y;
    float det = v1.x * v2.y - v1.y * v2.x;
    if (std::fabs(det) <= EPS) {
        auto sq = [](P2 a, P2 b) { float dx = a.x - b.x, dy = a.y - b.y; return dx * dx + dy * dy; };
        float d1 = sq(pts[0], pts[1]), d2 = sq(pts[0], pts[2]), d3 = sq(pts[1], pts[2]);
        radius = std::sqrt(std::max(d1, std::max(d2, d3))) * 0.5f + EPS;
        if (d1 >= d2 && d1 >= d3) c = P2{(pts[0].x + pts[1].x) * 0.5f, (pts[0].y + pts[1].y) * 0.5f};
        else if (d2 >= d1 && d2 >= d3) c = P2{(pts[0].x + pts[2].x) * 0.5f, (pts[0].y + pts[2].y) * 0.5f};
        else c = P2{(pts[1].x + pts[2].x) * 0.5f, (pts[1].y + pts[2].y) * 0.5f};
        return;
    }
    float cx = (c1 * v2.y - c2 * v1.y) / det, cy = (v1.x * c2 - v2.x * c1) / det;
    c.x = cx; c.y = cy;
    cx -= pts[0].x; cy -= pts[0].y;
    radius = (float)std::sqrt(cx * cx + cy * cy) + EPS;
}
static void third(const P2* pts, int i, int j, P2& c, float& radius) {
    c.x = (pts[j].x + pts[i].x) / 2.0f; c.y = (pts[j].y + pts[i].y) / 2.0f;
    float dx = pts[j].x - pts[i].x, dy = pts[j].y - pts[i].y;
    radius = (float)nrm(dx, dy) / 2.0f + EPS;
    for (int k = 0; k < j; ++k) {
        dx = c.x - pts[k].x; dy = c.y - pts[k].y;
        if (nrm(dx, dy) < radius) continue;
        P2 t[3] = {pts[i], pts[j], pts[k]}; P2 nc{0, 0}; float nr = 0;
        circle3(t, nc, nr);
        if (nr > 0) { radius = nr; c = nc; }
    }
}
static void second(const P2* pts, int i, P2& c, float& radius) {
    c.x = (pts[0].x + pts[i].x) / 2.0f; c.y = (pts[0].y + pts[i].y) / 2.0f;
    float dx = pts[0].x - pts[i].x, dy = pts[0].y - pts[i].y;
    radius = (float)nrm(dx, dy) / 2.0f + EPS;
    for (int j = 1; j < i; ++j) {
        dx = c.x - pts[j].x; dy = c.y - pts[j].y;
        if (nrm(dx, dy) < radius) continue;
        P2 nc{0, 0}; float nr = 0;
        third(pts, i, j, nc, nr);
        if (nr > 0) { radius = nr; c = nc; }
    }
}
}  // namespace mec

void min_enclosing_circle_f32(const float* xy, size_t n, float& cx, float& cy, float& r) {
    using namespace mec;
    cx = cy = 0.f; r = 0.f;
    if (n == 0) return;
    const P2* pts = reinterpret_cast<const P2*>(xy);
    if (n == 1) { cx = pts[0].x; cy = pts[0].y; r = EPS; return; }
    if (n == 2) {
        cx = (pts[0].x + pts[1].x) / 2.0f; cy = (pts[0].y + pts[1].y) / 2.0f;
        r = (float)(nrm(pts[0].x - pts[1].x, pts[0].y - pts[1].y) / 2.0) + EPS; return;
    }
    P2 c{(pts[0].x + pts[1].x) / 2.0f, (pts[0].y + pts[1].y) / 2.0f};
    float radius = (float)nrm(pts[0].x - pts[1].x, pts[0].y - pts[1].y) / 2.0f + EPS;
    for (int i = 2; i < (int)n; ++i) {
        float dx = pts[i].x - c.x, dy = pts[i].y - c.y;
        float d = (float)nrm(dx, dy);
        if (d < radius) continue;
        P2 nc{0, 0}; float nr = 0;
        second(pts, i, nc, nr);
        if (nr > 0) { radius = nr; c = nc; }
    }
    cx = c.x; cy = c.y; r = radius;
}

static void mec_i32(const int32_t* xy, size_t n, float& cx, float& cy, float& r) {
    std::vector<float> f(2 * n);
    for (size_t i = 0; i < 2 * n; i++) f[i] = (float)xy[i];
    min_enclosing_circle_f32(f.data(), n, cx, cy, r);
}

// ----------------------------------------------------------------------------------------------
// Stage 05: _scale_one (05:82-96): float32 (x*sx + y*0) + dx, truncation to int32.
// ----------------------------------------------------------------------------------------------
void scale_polys(const PolyList& in, float sx, float sy, float dx, float dy, PolyList& out) {
    out.clear();
    out.off = in.off;
    out.pts.resize(in.pts.size());
    for (size_t i = 0; i < in.pts.size() / 2; i++) {
        float x = (float)in.pts[2 * i] * sx; x = x + dx;
        float y = (float)in.pts[2 * i + 1] * sy; y = y + dy;
        out.pts[2 * i] = (int32_t)x; out.pts[2 * i + 1] = (int32_t)y;
    }
}

// ----------------------------------------------------------------------------------------------
// Greedy nearest-neighbour reorder with flips.
// ----------------------------------------------------------------------------------------------
static inline float d2_f32(int32_t ax, int32_t ay, int32_t bx, int32_t by) {
    float dx = (float)ax - (float)bx, dy = (float)ay - (float)by;
    float qx = dx * dx, qy = dy * dy;
    return qx + qy;
}

static void emit_poly(const PolyList& in, size_t i, bool flip, PolyList& out) {
    size_t n = in.npts(i); const int32_t* p = in.p(i);
    if (!flip) out.add(p, n);
    else { for (size_t k = n; k-- > 0;) out.push_pt(p[2 * k], p[2 * k + 1]); out.end_poly(); }
}

// 07:19-95 reorder_one_color
void sort_contours07(const PolyList& in, PolyList& out) {
    out.clear();
    size_t n = in.count();
    if (!n) return;
    std::vector<int32_t> sx(n), sy(n), ex(n), ey(n); std::vector<u8> closed(n), used(n, 0);
    std::vector<double> len(n);
    for (size_t i = 0; i < n; i++) {
        const int32_t* p = in.p(i); size_t m = in.npts(i);
        bool cl = p[0] == p[2 * (m - 1)] && p[1] == p[2 * (m - 1) + 1];
        size_t last = (cl && m > 1) ? m - 2 : m - 1;
        sx[i] = p[0]; sy[i] = p[1]; ex[i] = p[2 * last]; ey[i] = p[2 * last + 1]; closed[i] = cl;
        len[i] = arc_length_i32(p, m, true);
    }
    size_t cur = 0;
    for (size_t i = 1; i < n; i++) if (len[i] > len[cur]) cur = i;
    std::vector<size_t> order{cur}; std::vector<u8> flips{0}; used[cur] = 1;
    int32_t cx = closed[cur] ? sx[cur] : ex[cur], cy = closed[cur] ? sy[cur] : ey[cur];
    for (size_t step = 1; step < n; step++) {
        long best = -1; bool bflip = false; float bd = 0; bool have = false;
        for (size_t i = 0; i < n; i++) {
            if (used[i]) continue;
            float ds = d2_f32(sx[i], sy[i], cx, cy), de = d2_f32(ex[i], ey[i], cx, cy);
            if (closed[i]) { if (!have || ds < bd) { bd = ds; best = (long)i; bflip = false; have = true; } }
            else if (ds <= de) { if (!have || ds < bd) { bd = ds; best = (long)i; bflip = false; have = true; } }
            else { if (!have || de < bd) { bd = de; best = (long)i; bflip = true; have = true; } }
        }
        used[best] = 1; order.push_back((size_t)best); flips.push_back(bflip);
        if (closed[best]) { cx = sx[best]; cy = sy[best]; }
        else if (!bflip) { cx = ex[best]; cy = ey[best]; } else { cx = sx[best]; cy = sy[best]; }
    }
    for (size_t k = 0; k < n; k++) emit_poly(in, order[k], flips[k], out);
}

// 08:223-248 _reorder_only  /  10:69-97 _reorder_for_travel
void reorder_only(const PolyList& in, PolyList& out, int length_kind) {
    out.clear();
    size_t n = in.count();
    if (!n) return;
    std::vector<u8> used(n, 0);
    std::vector<double> len(n);
    for (size_t i = 0; i < n; i++)
        len[i] = length_kind == 0 ? (double)poly_perimeter_f32(in.p(i), in.npts(i)) : arc_length_i32(in.p(i), in.npts(i), false);
    size_t cur = 0;
    for (size_t i = 1; i < n; i++) if (len[i] > len[cur]) cur = i;
    auto S = [&](size_t i, int c) { return in.p(i)[c]; };
    auto E = [&](size_t i, int c) { return in.p(i)[2 * (in.npts(i) - 1) + c]; };
    std::vector<size_t> order{cur}; std::vector<u8> flips{0}; used[cur] = 1;
    int32_t cx = E(cur, 0), cy = E(cur, 1);
    for (size_t step = 1; step < n; step++) {
        long best = -1; bool bflip = false; float bd = 0;
        for (size_t i = 0; i < n; i++) {
            if (used[i]) continue;
            float ds = d2_f32(S(i, 0), S(i, 1), cx, cy), de = d2_f32(E(i, 0), E(i, 1), cx, cy);
            if (ds <= de) { if (best < 0 || ds < bd) { bd = ds; best = (long)i; bflip = false; } }
            else { if (best < 0 || de < bd) { bd = de; best = (long)i; bflip = true; } }
        }
        used[best] = 1; order.push_back((size_t)best); flips.push_back(bflip);
        if (bflip) { cx = S(best, 0); cy = S(best, 1); } else { cx = E(best, 0); cy = E(best, 1); }
    }
    for (size_t k = 0; k < n; k++) emit_poly(in, order[k], flips[k], out);
}

// ----------------------------------------------------------------------------------------------
// 08:53-64 _resample_arclen.  Input float32 points; `closed_in` = result of _is_closed on the
// caller's array.  Output float64 (x,y) pairs, or (is_f32_passthrough) the float32 input.
// ----------------------------------------------------------------------------------------------
void resample_arclen(const float* xy, size_t n, bool closed_in, double step, std::vector<double>& out, bool& pass) {
    out.clear(); pass = false;
    auto passthrough = [&](size_t m) { pass = true; out.resize(2 * m); for (size_t i = 0; i < 2 * m; i++) out[i] = xy[i]; };
    if (n < 2) { passthrough(n); return; }
    if (closed_in) n -= 1;
    std::vector<double> s(n);  // float32 cumsum values, held as float64 (np.concatenate with [0.0])
    s[0] = 0.0; float acc = 0.f;
    for (size_t i = 0; i + 1 < n; i++) {
        float dx = xy[2 * i + 2] - xy[2 * i], dy = xy[2 * i + 3] - xy[2 * i + 1];
        float q = dx * dx + dy * dy;   // (x*x) + (y*y) in float32
        float seg = std::sqrt(q);
        acc = (i == 0) ? seg : acc + seg;
        s[i + 1] = (double)acc;
    }
    double total = s[n - 1];
    if (total <= step) { passthrough(n); return; }
    // np.arange(0.0, total, step, dtype=float32)
    size_t m = (size_t)std::ceil(total / step);
    float t0 = 0.0f, t1 = (float)(0.0 + step); float delta = t1 - t0;
    out.resize(2 * m);
    size_t k = 0;
    for (size_t i = 0; i < m; i++) {
        float tf = (i == 0) ? t0 : (i == 1 ? t1 : t0 + (float)i * delta);
        double t = (double)tf;
        // searchsorted(s, t, 'right') - 1, clipped to [0, n-2]; t is non-decreasing so k only advances
        while (k + 1 < n && s[k + 1] <= t) k++;
        size_t kk = std::min(k, n - 2);
        double u = (t - s[kk]) / std::max(1e-6, s[kk + 1] - s[kk]);
        double a = 1.0 - u;
        out[2 * i] = (double)xy[2 * kk] * a + (double)xy[2 * kk + 2] * u;
        out[2 * i + 1] = (double)xy[2 * kk + 1] * a + (double)xy[2 * kk + 3] * u;
    }
}

// ----------------------------------------------------------------------------------------------
// 08:68-99 _PointHash
// ----------------------------------------------------------------------------------------------
struct PointHash {
    double r, cell, inv;
    std::unordered_map<int64_t, std::vector<double>> g;
    PointHash(double radius, double c) : r(radius) { cell = (c > 0) ? c : std::max(4.0, radius); inv = 1.0 / cell; }
    static int64_t key(int64_t cx, int64_t cy) { return (cx << 32) ^ (cy & 0xffffffffLL); }
    void cellof(double x, double y, int64_t& cx, int64_t& cy) const { cx = (int64_t)std::floor(x * inv); cy = (int64_t)std::floor(y * inv); }
    bool near(double x, double y) const {
        double R2 = r * r; int64_t cx, cy; cellof(x, y, cx, cy);
        for (int dx = -1; dx <= 1; dx++)
            for (int dy = -1; dy <= 1; dy++) {
                auto it = g.find(key(cx + dx, cy + dy));
                if (it == g.end()) continue;
                const std::vector<double>& a = it->second;
                for (size_t i = 0; i < a.size(); i += 2) {
                    double ddx = a[i] - x, ddy = a[i + 1] - y;
                    double qx = ddx * ddx, qy = ddy * ddy;
                    if (qx + qy <= R2) return true;
                }
            }
        return false;
    }
    void add(double x, double y) { int64_t cx, cy; cellof(x, y, cx, cy); auto& v = g[key(cx, cy)]; v.push_back(x); v.push_back(y); }
};

// float64 2-vector norm as evaluated by np.linalg.norm(np.array(a)-np.array(b)) in the reference
// environment: sqrt(ddot(v,v)); the OpenBLAS ddot kernel contracts to fma(dy,dy,dx*dx) (checked
// against numpy 2.2.6 / OpenBLAS 0.3.29 in the build container: 100000/100000 agree).
static inline double norm2_f64(double dx, double dy) { return std::sqrt(std::fma(dy, dy, dx * dx)); }

static void flush_seg(std::vector<double>& cur, PolyList& segs) {
    if (cur.size() >= 4) {
        for (size_t i = 0; i < cur.size(); i += 2) segs.push_pt((int32_t)cur[i], (int32_t)cur[i + 1]);
        segs.end_poly();
    }
    cur.clear();
}

// 08:117-183 _virtual_draw_split_with_mask_and_tail
void virtual_draw08(const int32_t* xy, size_t n_in, const Params08& P, u8* mask, PolyList& segs) {
    // _ensure_open
    size_t n = n_in;
    if (n >= 2 && xy[0] == xy[2 * (n - 1)] && xy[1] == xy[2 * (n - 1) + 1]) n -= 1;
    if (n < 2) return;
    std::vector<float> p(2 * n);
    for (size_t i = 0; i < 2 * n; i++) p[i] = (float)xy[i];
    bool closed = n > 2 && p[0] == p[2 * (n - 1)] && p[1] == p[2 * (n - 1) + 1];
    std::vector<double> S; bool pass;
    resample_arclen(p.data(), n, closed, std::max(1.0, P.sample_step), S, pass);
    size_t m = S.size() / 2;
    if (m < 2) return;
    const int W = P.W, H = P.H; const int rad = P.brush_forbid / 2;
    PointHash hash(P.col_rad, P.grid_stride);
    std::deque<std::pair<double, double>> tail; double tail_len = 0.0;
    bool has_last = false; int lx = 0, ly = 0;
    std::vector<double> cur;
    auto stamp_old = [&](double ox, double oy) {
        int64_t xi = round_half_even(ox), yi = round_half_even(oy);
        if (xi >= 0 && xi < W && yi >= 0 && yi < H) {
            if (has_last) stamp_capsule(mask, H, W, lx, ly, (int)xi, (int)yi, rad, 255);
            has_last = true; lx = (int)xi; ly = (int)yi;
        }
    };
    auto pop_old = [&]() {
        while (!tail.empty() && tail_len > P.tail_len_px) {
            auto o = tail.front(); tail.pop_front();
            hash.add(o.first, o.second);
            if (!tail.empty()) tail_len -= norm2_f64(tail.front().first - o.first, tail.front().second - o.second);
            else tail_len = 0.0;
            stamp_old(o.first, o.second);
        }
    };
    for (size_t i = 0; i < m; i++) {
        double x = S[2 * i], y = S[2 * i + 1];
        if (!tail.empty()) tail_len += norm2_f64(x - tail.back().first, y - tail.back().second);
        tail.emplace_back(x, y);
        pop_old();
        int64_t xi = round_half_even(x), yi = round_half_even(y);
        if (xi < 0 || yi < 0 || xi >= W || yi >= H) { flush_seg(cur, segs); continue; }
        if (mask[(size_t)yi * W + xi] == 255 || hash.near(x, y)) { flush_seg(cur, segs); continue; }
        cur.push_back(x); cur.push_back(y);
    }
    pop_old();
    while (!tail.empty()) { auto o = tail.front(); tail.pop_front(); stamp_old(o.first, o.second); }
    flush_seg(cur, segs);
}

// 08:185-196
void split_on_long_jumps08(const int32_t* xy, size_t n, double max_jump, PolyList& out) {
    if (n < 2) return;
    out.push_pt(xy[0], xy[1]);
    for (size_t i = 1; i < n; i++) {
        float dx = (float)xy[2 * i] - (float)xy[2 * i - 2], dy = (float)xy[2 * i + 1] - (float)xy[2 * i - 1];
        float d = std::sqrt(dx * dx + dy * dy);
        if ((double)d > max_jump && out.open_len() >= 2) { out.end_poly(); out.push_pt(xy[2 * i], xy[2 * i + 1]); }
        else out.push_pt(xy[2 * i], xy[2 * i + 1]);
    }
    if (out.open_len() >= 2) out.end_poly(); else out.abort_poly();
}

// 10:49-63
void split_on_long_jumps10(const int32_t* xy, size_t n, double max_jump, PolyList& out) {
    if (n < 2) return;
    out.push_pt(xy[0], xy[1]);
    for (size_t i = 1; i < n; i++) {
        double dx = (double)((float)xy[2 * i] - (float)xy[2 * i - 2]), dy = (double)((float)xy[2 * i + 1] - (float)xy[2 * i - 1]);
        if (std::hypot(dx, dy) > max_jump) { if (out.open_len() >= 2) out.end_poly(); else out.abort_poly(); }
        out.push_pt(xy[2 * i], xy[2 * i + 1]);
    }
    if (out.open_len() >= 2) out.end_poly(); else out.abort_poly();
}

// 08:198-216
void split_small_and_taps08(const PolyList& in, const Params08& P, PolyList& kept, Taps& taps) {
    for (size_t i = 0; i < in.count(); i++) {
        const int32_t* p = in.p(i); size_t n = in.npts(i);
        if (n < 2) continue;
        int32_t x0 = p[0], x1 = p[0], y0 = p[1], y1 = p[1];
        for (size_t k = 1; k < n; k++) { x0 = std::min(x0, p[2 * k]); x1 = std::max(x1, p[2 * k]); y0 = std::min(y0, p[2 * k + 1]); y1 = std::max(y1, p[2 * k + 1]); }
        double d = (double)std::max(x1 - x0, y1 - y0);
        if (d <= P.tap_diam && d <= P.tap_max_dim) {
            double per = (double)poly_perimeter_f32(p, n);
            if (per <= P.tap_max_per && (int)n <= P.tap_max_v) {
                float cx, cy, r; mec_i32(p, n, cx, cy, r);
                taps.add((int32_t)round_half_even((double)cx), (int32_t)round_half_even((double)cy));
                continue;
            }
        }
        if (d < P.min_keep) continue;
        size_t m = n;
        if (m >= 2 && p[0] == p[2 * (m - 1)] && p[1] == p[2 * (m - 1) + 1]) m -= 1;  // _ensure_open
        kept.add(p, m);
    }
}

// 08:319-338.  bboxes: 4 ints per line (x0,y0,x1,y1).  Groups ordered by smallest member index.
void cluster_by_overlap(const std::vector<int32_t>& b, std::vector<std::vector<int>>& groups) {
    groups.clear();
    int n = (int)(b.size() / 4);
    if (!n) return;
    std::vector<int> parent(n);
    for (int i = 0; i < n; i++) parent[i] = i;
    auto find = [&](int x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++) {
            bool ov = !(b[4 * i + 2] < b[4 * j] || b[4 * j + 2] < b[4 * i] || b[4 * i + 3] < b[4 * j + 1] || b[4 * j + 3] < b[4 * i + 1]);
            if (ov) { int ra = find(i), rb = find(j); if (ra != rb) parent[rb] = ra; }
        }
    std::map<int, int> slot;
    for (int i = 0; i < n; i++) {
        int r = find(i);
        auto it = slot.find(r);
        if (it == slot.end()) { slot[r] = (int)groups.size(); groups.push_back({i}); }
        else groups[it->second].push_back(i);
    }
}

static const int OFFS8[8][2] = {{-1, -1}, {-1, 0}, {-1, 1}, {0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1}};  // (dy,dx) 08:252

// 08:261-280
void bfs_path(const u8* img, int h, int w, int sy, int sx, int gy, int gx, std::vector<int32_t>& path) {
    path.clear();
    if (sy == gy && sx == gx) { path = {sy, sx}; return; }
    std::vector<int32_t> prev((size_t)h * w, -1); std::vector<u8> seen((size_t)h * w, 0);
    std::vector<int32_t> que{sy * w + sx}; size_t head = 0; seen[(size_t)sy * w + sx] = 1;
    int goal = gy * w + gx;
    while (head < que.size()) {
        int c = que[head++];
        if (c == goal) break;
        int y = c / w, x = c % w;
        for (auto& o : OFFS8) {
            int ny = y + o[0], nx = x + o[1];
            if (ny < 0 || ny >= h || nx < 0 || nx >= w) continue;
            size_t j = (size_t)ny * w + nx;
            if (!img[j] || seen[j]) continue;
            seen[j] = 1; prev[j] = c; que.push_back((int32_t)j);
        }
    }
    if (prev[goal] == -1) return;
    std::vector<int32_t> rev{goal};
    int c = goal, start = sy * w + sx;
    while (c != start) { int pc = prev[c]; if (pc == -1) { path.clear(); return; } rev.push_back(pc); c = pc; }
    for (size_t i = rev.size(); i-- > 0;) { path.push_back(rev[i] / w); path.push_back(rev[i] % w); }
}

// 08:282-293
static int farthest(const u8* img, int h, int w, int src) {
    std::vector<u8> seen((size_t)h * w, 0);
    std::vector<int32_t> que{src}; size_t head = 0; seen[src] = 1; int last = src;
    while (head < que.size()) {
        int c = que[head++]; last = c;
        int y = c / w, x = c % w;
        for (auto& o : OFFS8) {
            int ny = y + o[0], nx = x + o[1];
            if (ny < 0 || ny >= h || nx < 0 || nx >= w) continue;
            size_t j = (size_t)ny * w + nx;
            if (!img[j] || seen[j]) continue;
            seen[j] = 1; que.push_back((int32_t)j);
        }
    }
    return last;
}

// 08:295-317
void component_best_path(const u8* comp, int h, int w, bool has_a, int ay, int ax, bool has_b, int by, int bx,
                         int min_len, std::vector<int32_t>& path) {
    path.clear();
    int seed = -1;
    for (size_t i = 0; i < (size_t)h * w; i++) if (comp[i]) { seed = (int)i; break; }
    if (seed < 0) return;
    size_t need = (size_t)std::max(2, min_len);
    if (has_a && has_b) {
        if (ay >= 0 && ay < h && ax >= 0 && ax < w && by >= 0 && by < h && bx >= 0 && bx < w &&
            comp[(size_t)ay * w + ax] && comp[(size_t)by * w + bx]) {
            bfs_path(comp, h, w, ay, ax, by, bx, path);
            if (path.size() / 2 >= need) return;
        }
    }
    int u = farthest(comp, h, w, seed), v = farthest(comp, h, w, u);
    bfs_path(comp, h, w, u / w, u % w, v / w, v % w, path);
    if (path.size() / 2 < need) path.clear();
}

// 08:376-469
void post_skeleton_merge(const PolyList& lines, const Params08& P, PolyList& merged) {
    merged.clear();
    size_t n = lines.count();
    if (!n) return;
    int exp = P.post_brush * 2 + 6;
    std::vector<int32_t> bxs(4 * n);
    for (size_t i = 0; i < n; i++) {
        const int32_t* p = lines.p(i); size_t m = lines.npts(i);
        int32_t x0 = p[0], x1 = p[0], y0 = p[1], y1 = p[1];
        for (size_t k = 1; k < m; k++) { x0 = std::min(x0, p[2 * k]); x1 = std::max(x1, p[2 * k]); y0 = std::min(y0, p[2 * k + 1]); y1 = std::max(y1, p[2 * k + 1]); }
        bxs[4 * i] = x0 - exp; bxs[4 * i + 1] = y0 - exp; bxs[4 * i + 2] = x1 + exp; bxs[4 * i + 3] = y1 + exp;
    }
    std::vector<std::vector<int>> groups;
    cluster_by_overlap(bxs, groups);
    std::vector<float> per(n);
    for (size_t i = 0; i < n; i++) per[i] = poly_perimeter_f32(lines.p(i), lines.npts(i));
    for (auto& idxs : groups) {
        int longest = idxs[0];
        for (int j : idxs) if (per[j] > per[longest]) longest = j;
        const int32_t* lp = lines.p(longest); size_t lm = lines.npts(longest);
        int a0x = lp[0], a0y = lp[1], a1x = lp[2 * (lm - 1)], a1y = lp[2 * (lm - 1) + 1];
        int x0 = bxs[4 * idxs[0]], y0 = bxs[4 * idxs[0] + 1], x1 = bxs[4 * idxs[0] + 2], y1 = bxs[4 * idxs[0] + 3];
        for (int j : idxs) { x0 = std::min(x0, bxs[4 * j]); y0 = std::min(y0, bxs[4 * j + 1]); x1 = std::max(x1, bxs[4 * j + 2]); y1 = std::max(y1, bxs[4 * j + 3]); }
        int w = std::max(1, x1 - x0), h = std::max(1, y1 - y0);
        std::vector<u8> roi((size_t)h * w, 0), sk((size_t)h * w);
        for (int j : idxs) if (lines.npts(j) >= 2) stamp_polyline(roi.data(), h, w, lines.p(j), lines.npts(j), x0, y0, std::max(1, P.post_brush) / 2);
        zhang_suen_std(roi.data(), sk.data(), h, w, 48);
        bool any = false;
        for (u8 v : sk) if (v) { any = true; break; }
        if (!any) continue;
        std::vector<int32_t> lab((size_t)h * w);
        int num = ccl8(sk.data(), lab.data(), h, w);
        std::vector<int32_t> skpix;   // skeleton pixels in raster order
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) if (sk[(size_t)y * w + x]) { skpix.push_back(y); skpix.push_back(x); }
        auto nearest = [&](int xa, int ya, int& oy, int& ox) {
            int64_t best = INT64_MAX; oy = ox = -1;
            for (size_t i = 0; i < skpix.size(); i += 2) {
                int64_t dy = (int64_t)skpix[i] - (ya - y0), dx = (int64_t)skpix[i + 1] - (xa - x0), d = dy * dy + dx * dx;
                if (d < best) { best = d; oy = skpix[i]; ox = skpix[i + 1]; }
            }
        };
        int a0y_, a0x_, a1y_, a1x_;
        nearest(a0x, a0y, a0y_, a0x_); nearest(a1x, a1y, a1y_, a1x_);
        // per-component bounding boxes, so that each component is examined on its own crop (same result as the
        // reference's full-ROI `lab == cc` image: BFS order and the raster-first seed are invariant under cropping)
        std::vector<int> cbx0(num + 1, w), cby0(num + 1, h), cbx1(num + 1, -1), cby1(num + 1, -1);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int l = lab[(size_t)y * w + x]; if (!l) continue;
                cbx0[l] = std::min(cbx0[l], x); cbx1[l] = std::max(cbx1[l], x); cby0[l] = std::min(cby0[l], y); cby1[l] = std::max(cby1[l], y);
            }
        std::vector<u8> comp;
        std::vector<int32_t> path;
        for (int cc = 1; cc <= num; cc++) {
            int ox = cbx0[cc], oy = cby0[cc], cw = cbx1[cc] - cbx0[cc] + 1, chh = cby1[cc] - cby0[cc] + 1;
            comp.assign((size_t)cw * chh, 0);
            for (int y = 0; y < chh; y++) for (int x = 0; x < cw; x++) comp[(size_t)y * cw + x] = lab[(size_t)(y + oy) * w + x + ox] == cc ? 255 : 0;
            auto inside = [&](int yy, int xx) { return yy >= oy && yy < oy + chh && xx >= ox && xx < ox + cw && comp[(size_t)(yy - oy) * cw + (xx - ox)]; };
            bool ha = a0y_ >= 0 && inside(a0y_, a0x_), hb = a1y_ >= 0 && inside(a1y_, a1x_);
            component_best_path(comp.data(), chh, cw, ha, a0y_ - oy, a0x_ - ox, hb, a1y_ - oy, a1x_ - ox, P.post_minlen, path);
            for (size_t i = 0; i < path.size(); i += 2) { path[i] += oy; path[i + 1] += ox; }
            size_t pl = path.size() / 2;
            if (pl < 2) continue;
            std::vector<float> arr(2 * pl);
            for (size_t i = 0; i < pl; i++) { arr[2 * i] = (float)(x0 + path[2 * i + 1]); arr[2 * i + 1] = (float)(y0 + path[2 * i]); }
            bool closed = pl > 2 && arr[0] == arr[2 * (pl - 1)] && arr[1] == arr[2 * (pl - 1) + 1];
            std::vector<double> rs; bool pass;
            resample_arclen(arr.data(), pl, closed, P.post_step, rs, pass);
            size_t m = rs.size() / 2;
            if (m < 2) continue;
            // RDP (08:452-462) on float32 points, explicit LIFO stack
            std::vector<float> Pf(2 * m);
            for (size_t i = 0; i < 2 * m; i++) Pf[i] = (float)rs[i];
            std::vector<u8> keep(m, 0); keep[0] = keep[m - 1] = 1;
            std::vector<std::pair<size_t, size_t>> stack{{0, m - 1}};
            while (!stack.empty()) {
                auto se = stack.back(); stack.pop_back();
                size_t s = se.first, e = se.second;
                if (e <= s + 1) continue;
                float ax = Pf[2 * s], ay = Pf[2 * s + 1], bx = Pf[2 * e], by = Pf[2 * e + 1];
                float segx = bx - ax, segy = by - ay, nx = -segy, ny = segx;
                float q = segx * segx + segy * segy;
                double seg_len = (double)std::sqrt(q) + 1e-12;
                float seg_len_f = (float)seg_len;   // numpy 2 weak-scalar promotion: float32 array / python float
                float bestd = -1.f; size_t bi = 0;
                for (size_t i = s + 1; i < e; i++) {
                    float dx = Pf[2 * i] - ax, dy = Pf[2 * i + 1] - ay;
                    float t0 = dx * nx, t1 = dy * ny;
                    float d = std::fabs(t0 + t1) / seg_len_f;
                    if (d > bestd) { bestd = d; bi = i; }
                }
                if (bestd > (float)P.post_eps) { keep[bi] = 1; stack.push_back({s, bi}); stack.push_back({bi, e}); }
            }
            for (size_t i = 0; i < m; i++) if (keep[i]) merged.push_pt((int32_t)Pf[2 * i], (int32_t)Pf[2 * i + 1]);
            merged.end_poly();
        }
    }
}

// 08:484-557 process_layer
void stage08_layer(const PolyList& sorted, const Params08& P, PolyList& lines, Taps& taps) {
    lines.clear(); taps.xy.clear();
    if (!sorted.count()) return;
    PolyList kept; split_small_and_taps08(sorted, P, kept, taps);
    size_t total = kept.count();
    PolyList lines2; Taps taps2;
    if (total) {
        std::vector<float> per(total);
        for (size_t i = 0; i < total; i++) per[i] = poly_perimeter_f32(kept.p(i), kept.npts(i));
        std::vector<size_t> order(total);
        for (size_t i = 0; i < total; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return per[a] > per[b]; });
        std::vector<u8> forbid((size_t)P.W * P.H, 0);
        PolyList cleaned;
        for (size_t i : order) {
            PolyList segs; virtual_draw08(kept.p(i), kept.npts(i), P, forbid.data(), segs);
            for (size_t s = 0; s < segs.count(); s++) {
                size_t before = cleaned.count();
                split_on_long_jumps08(segs.p(s), segs.npts(s), P.max_jump, cleaned);
                if (cleaned.count() == before) cleaned.add(segs.p(s), segs.npts(s));
            }
        }
        split_small_and_taps08(cleaned, P, lines2, taps2);
        taps.xy.insert(taps.xy.end(), taps2.xy.begin(), taps2.xy.end());
    }
    PolyList merged;
    if (P.post_on && lines2.count() > 0) post_skeleton_merge(lines2, P, merged); else merged = lines2;
    reorder_only(merged, lines, 0);
}

// ----------------------------------------------------------------------------------------------
// Stage 10
// ----------------------------------------------------------------------------------------------
static void flush10(std::vector<float>& cur, PolyList& out) {
    if (cur.size() >= 4) { for (size_t i = 0; i < cur.size(); i += 2) out.push_pt((int32_t)cur[i], (int32_t)cur[i + 1]); out.end_poly(); }
    cur.clear();
}

// 10:142-177
void cut_poly_against_mask(const int32_t* xy, size_t n, const u8* forb, int H, int W, double step_px, PolyList& out) {
    if (n < 2) return;
    auto blocked = [&](float x, float y) {
        int64_t xi = round_half_even((double)x), yi = round_half_even((double)y);
        return yi >= 0 && yi < H && xi >= 0 && xi < W && forb[(size_t)yi * W + xi] != 0;
    };
    std::vector<float> cur;
    if (!blocked((float)xy[0], (float)xy[1])) { cur.push_back((float)xy[0]); cur.push_back((float)xy[1]); }
    for (size_t i = 1; i < n; i++) {
        float p0x = (float)xy[2 * i - 2], p0y = (float)xy[2 * i - 1], p1x = (float)xy[2 * i], p1y = (float)xy[2 * i + 1];
        float vx = p1x - p0x, vy = p1y - p0y;
        double L = (double)hypot_f32(vx, vy);
        if (L <= 1e-6) continue;
        long nn = std::max(1L, (long)std::ceil(L / std::max(1.0, step_px)));
        for (long k = 1; k <= nn; k++) {
            float t = (float)((double)k / (double)nn);
            float qx = vx * t; qx = p0x + qx;
            float qy = vy * t; qy = p0y + qy;
            if (blocked(qx, qy)) flush10(cur, out);
            else { cur.push_back(qx); cur.push_back(qy); }
        }
    }
    flush10(cur, out);
}

// 10:99-118
void tiny_and_taps10(const PolyList& in, const Params10& P, PolyList& kept, Taps& taps) {
    for (size_t i = 0; i < in.count(); i++) {
        const int32_t* p = in.p(i); size_t n = in.npts(i);
        float cx, cy, r; mec_i32(p, n, cx, cy, r);
        double d = 2.0 * (double)r;
        if (d <= P.tap_diam) {
            double per = arc_length_i32(p, n, false);
            if (per <= P.tap_max_per && (int)n <= P.tap_max_v) {
                taps.add((int32_t)round_half_even((double)cx), (int32_t)round_half_even((double)cy));
                continue;
            }
        }
        if (d >= P.min_keep) kept.add(p, n);
    }
}

// 10:236-267, one layer against the cumulative mask
void stage10_layer(const PolyList& lines_in, const Taps& taps_in, const Params10& P, u8* forbidden,
                   PolyList& lines_out, Taps& taps_out) {
    lines_out.clear(); taps_out.xy.clear();
    PolyList cut;
    for (size_t i = 0; i < lines_in.count(); i++)
        cut_poly_against_mask(lines_in.p(i), lines_in.npts(i), forbidden, P.H, P.W, P.step_px, cut);
    PolyList cut2;
    for (size_t i = 0; i < cut.count(); i++) {
        size_t before = cut2.count();
        split_on_long_jumps10(cut.p(i), cut.npts(i), P.max_jump, cut2);
        if (cut2.count() == before) cut2.add(cut.p(i), cut.npts(i));
    }
    PolyList keep; Taps from_lines;
    tiny_and_taps10(cut2, P, keep, from_lines);
    reorder_only(keep, lines_out, 1);
    int thickness = (int)std::max<int64_t>(1, round_half_even(P.D_lines));
    for (size_t i = 0; i < lines_out.count(); i++)
        if (lines_out.npts(i) >= 2) stamp_polyline(forbidden, P.H, P.W, lines_out.p(i), lines_out.npts(i), 0, 0, thickness / 2);
    int radius = (int)std::max<int64_t>(1, round_half_even(P.D_taps / 2.0));
    std::vector<int32_t> seq(taps_in.xy);
    seq.insert(seq.end(), from_lines.xy.begin(), from_lines.xy.end());
    for (size_t i = 0; i < seq.size() / 2; i++) {
        int x = seq[2 * i], y = seq[2 * i + 1];
        bool blocked = y >= 0 && y < P.H && x >= 0 && x < P.W && forbidden[(size_t)y * P.W + x] != 0;
        if (!blocked) { taps_out.add(x, y); stamp_disc(forbidden, P.H, P.W, x, y, radius, 255); }
    }
}

// ----------------------------------------------------------------------------------------------
// Stage 12: _build_ops_for_layer (12:85-187)
// ----------------------------------------------------------------------------------------------
void build_ops12(const PolyList& lines, const Taps& taps, double R, std::vector<Op>& ops) {
    ops.clear();
    struct LC { int i; double sx, sy, ex, ey; float len; };
    struct TC { double x, y; };
    std::vector<LC> L; std::vector<TC> T;
    for (size_t i = 0; i < lines.count(); i++) {
        size_t n = lines.npts(i); if (n < 2) continue;
        const int32_t* p = lines.p(i);
        L.push_back({(int)i, (double)(float)p[0], (double)(float)p[1], (double)(float)p[2 * (n - 1)], (double)(float)p[2 * (n - 1) + 1], poly_len12_f32(p, n)});
    }
    for (size_t i = 0; i < taps.count(); i++) T.push_back({(double)taps.xy[2 * i], (double)taps.xy[2 * i + 1]});
    if (L.empty() && T.empty()) return;
    double px = 0, py = 0;
    auto dist = [&](double x, double y) { return std::hypot(px - x, py - y); };
    auto drain = [&]() {
        std::vector<TC> kept;
        for (auto& t : T) {
            if (dist(t.x, t.y) <= R) { ops.push_back({1, -1, 0, (int)round_half_even(t.x), (int)round_half_even(t.y)}); px = t.x; py = t.y; }
            else kept.push_back(t);
        }
        T.swap(kept);
    };
    if (!L.empty()) {
        size_t s = 0;
        for (size_t k = 1; k < L.size(); k++) if (L[k].len > L[s].len) s = k;
        LC f = L[s]; L.erase(L.begin() + (long)s);
        int flip = 0;
        if (dist(f.ex, f.ey) < dist(f.sx, f.sy)) { flip = 1; std::swap(f.sx, f.ex); std::swap(f.sy, f.ey); }
        ops.push_back({0, f.i, flip, 0, 0});
        px = f.ex; py = f.ey;
        drain();
    } else {
        size_t s = 0; double bd = dist(T[0].x, T[0].y);
        for (size_t k = 1; k < T.size(); k++) { double d = dist(T[k].x, T[k].y); if (d < bd) { bd = d; s = k; } }
        TC f = T[s]; T.erase(T.begin() + (long)s);
        ops.push_back({1, -1, 0, (int)round_half_even(f.x), (int)round_half_even(f.y)});
        px = f.x; py = f.y;
    }
    while (!L.empty() || !T.empty()) {
        int kind = -1; long bi = -1; double bc = 1e20; int bflip = 0;
        for (size_t k = 0; k < L.size(); k++) {
            double d1 = dist(L[k].sx, L[k].sy), d2 = dist(L[k].ex, L[k].ey);
            if (d1 < bc) { bc = d1; kind = 0; bi = (long)k; bflip = 0; }
            if (d2 < bc) { bc = d2; kind = 0; bi = (long)k; bflip = 1; }
        }
        for (size_t k = 0; k < T.size(); k++) {
            double d = dist(T[k].x, T[k].y);
            if (d < bc) { bc = d; kind = 1; bi = (long)k; bflip = 0; }
        }
        if (kind == 0) {
            LC c = L[(size_t)bi]; L.erase(L.begin() + bi);
            ops.push_back({0, c.i, bflip, 0, 0});
            if (bflip) { px = c.sx; py = c.sy; } else { px = c.ex; py = c.ey; }
            drain();
        } else {
            TC c = T[(size_t)bi]; T.erase(T.begin() + bi);
            ops.push_back({1, -1, 0, (int)round_half_even(c.x), (int)round_half_even(c.y)});
            px = c.x; py = c.y;
        }
    }
}

}  // namespace orc
