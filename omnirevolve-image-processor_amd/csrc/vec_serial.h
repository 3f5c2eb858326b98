// csrc/vec_serial.h -- serial per-polyline / per-component routines of the vector stages (05..12), written as
// host+device functions so that the HIP kernels (one lane per polyline / component) and the CPU harness under
// tests/host/ compile the very same code.  dtype and rounding rules follow SURVEY App. A.5; OpenCV primitives
// (minEnclosingCircle, arcLength, thick lines as capsules) follow SURVEY App. B.8-B.9.
#pragma once
#include <cstdint>
#include <cmath>
#if defined(__HIPCC__)
#define ORIP_HD __host__ __device__
#else
#define ORIP_HD
#endif

namespace vs {

// ---- exact float helpers (no contraction: the TU is built with -ffp-contract=off) ----
// Points come through a getter pt(i) -> IPt, so the same arithmetic runs over an int32 x,y array (XYPtr) and over the cursor of a
// walk-coded list (vsrc.h).
struct IPt { int32_t x, y; };
struct XYPtr { const int32_t* xy; ORIP_HD IPt operator()(int64_t i) const { return IPt{xy[2 * i], xy[2 * i + 1]}; } };
template <class PT> ORIP_HD inline float seg_len_f32_p(const PT& pt, int64_t i) {       // np.linalg.norm(p[i+1]-p[i]) in float32 (08:25-28)
    const IPt a = pt(i), b = pt(i + 1);
    float dx = (float)b.x - (float)a.x, dy = (float)b.y - (float)a.y;
    float qx = dx * dx, qy = dy * dy;
    return sqrtf(qx + qy);
}
template <class PT> ORIP_HD inline float seg_hypot_f32_p(const PT& pt, int64_t i) {     // np.hypot on float32 (12:71-76), correctly rounded
    const IPt a = pt(i), b = pt(i + 1);
    float dx = (float)b.x - (float)a.x, dy = (float)b.y - (float)a.y;
    return (float)sqrt((double)dx * (double)dx + (double)dy * (double)dy);
}
ORIP_HD inline float seg_len_f32(const int32_t* xy, int64_t i) { return seg_len_f32_p(XYPtr{xy}, i); }
ORIP_HD inline float seg_hypot_f32(const int32_t* xy, int64_t i) { return seg_hypot_f32_p(XYPtr{xy}, i); }
ORIP_HD inline long long round_half_even(double v) { return (long long)rint(v); }

// numpy float32 pairwise summation of the n segment lengths of a polyline (ndarray.sum()); KIND 0: seg_len_f32, 1: seg_hypot_f32
template <int KIND, class PT>
ORIP_HD inline float pairwise_leaf_p(const PT& pt, int64_t s, int64_t n) {
    auto el = [&](int64_t i) { return KIND == 0 ? seg_len_f32_p(pt, s + i) : seg_hypot_f32_p(pt, s + i); };
    if (n < 8) { float r = 0.f; for (int64_t i = 0; i < n; i++) r += el(i); return r; }
    float r0 = el(0), r1 = el(1), r2 = el(2), r3 = el(3), r4 = el(4), r5 = el(5), r6 = el(6), r7 = el(7);
    int64_t i;
    for (i = 8; i < n - (n % 8); i += 8) { r0 += el(i); r1 += el(i + 1); r2 += el(i + 2); r3 += el(i + 3); r4 += el(i + 4); r5 += el(i + 5); r6 += el(i + 6); r7 += el(i + 7); }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += el(i);
    return res;
}
template <int KIND, class PT>
ORIP_HD inline float pairwise_seglen_sum_p(const PT& pt, int64_t npts) {
    int64_t n = npts - 1;
    if (n <= 0) return 0.f;
    if (n <= 128) return pairwise_leaf_p<KIND>(pt, 0, n);
    // explicit-stack evaluation of  pw(s,n) = n<=128 ? leaf : pw(s,n2) + pw(s+n2,n-n2),  n2 = n/2 - (n/2)%8
    int64_t fs[48], fn[48]; int fstate[48]; float fleft[48];
    int sp = 0; fs[0] = 0; fn[0] = n; fstate[0] = 0; sp = 1;
    float ret = 0.f;
    while (sp > 0) {
        int t = sp - 1;
        if (fn[t] <= 128) { ret = pairwise_leaf_p<KIND>(pt, fs[t], fn[t]); sp--; continue; }
        int64_t n2 = fn[t] / 2; n2 -= n2 % 8;
        if (fstate[t] == 0) { fstate[t] = 1; fs[sp] = fs[t]; fn[sp] = n2; fstate[sp] = 0; sp++; }
        else if (fstate[t] == 1) { fleft[t] = ret; fstate[t] = 2; fs[sp] = fs[t] + n2; fn[sp] = fn[t] - n2; fstate[sp] = 0; sp++; }
        else { ret = fleft[t] + ret; sp--; }
    }
    return ret;
}
template <int KIND> ORIP_HD inline float pairwise_leaf(const int32_t* xy, int64_t s, int64_t n) { return pairwise_leaf_p<KIND>(XYPtr{xy}, s, n); }
template <int KIND> ORIP_HD inline float pairwise_seglen_sum(const int32_t* xy, int64_t npts) { return pairwise_seglen_sum_p<KIND>(XYPtr{xy}, npts); }

// cv::arcLength on int points (07:50 closed, 10:43 open): float per-edge sqrt accumulated in double
template <class PT>
ORIP_HD inline double arc_length_p(const PT& pt, int64_t n, bool closed) {
    if (n <= 1) return 0.0;
    int64_t last = closed ? n - 1 : 0;
    const IPt pl = pt(last);
    float pvx = (float)pl.x, pvy = (float)pl.y;
    double per = 0;
    for (int64_t i = 0; i < n; i++) {
        const IPt q = pt(i);
        float x = (float)q.x, y = (float)q.y;
        float dx = x - pvx, dy = y - pvy;
        float qx = dx * dx, qy = dy * dy;
        per += (double)sqrtf(qx + qy);
        pvx = x; pvy = y;
    }
    return per;
}
ORIP_HD inline double arc_length(const int32_t* xy, int64_t n, bool closed) { return arc_length_p(XYPtr{xy}, n, closed); }

// ---- cv::minEnclosingCircle on int points converted to float (08:212, 10:46,113); recalled from OpenCV 4.x ----
struct P2 { float x, y; };
ORIP_HD inline P2 ipt(const int32_t* xy, int i) { return P2{(float)xy[2 * i], (float)xy[2 * i + 1]}; }
ORIP_HD inline double nrm2(float dx, float dy) { return sqrt((double)dx * dx + (double)dy * dy); }
ORIP_HD inline void mec_circle3(P2 p0, P2 p1, P2 p2, P2& c, float& radius) {
    const float EPS = 1.0e-4f;
    P2 v1{p1.x - p0.x, p1.y - p0.y}, v2{p2.x - p0.x, p2.y - p0.y};
    P2 m1{(p0.x + p1.x) / 2.0f, (p0.y + p1.y) / 2.0f};
    float a1 = m1.x * v1.x, b1 = m1.y * v1.y; float c1 = a1 + b1;
    P2 m2{(p0.x + p2.x) / 2.0f, (p0.y + p2.y) / 2.0f};
    float a2 = m2.x * v2.x, b2 = m2.y * v2.y; float c2 = a2 + b2;
    float d1_ = v1.x * v2.y, d2_ = v1.y * v2.x; float det = d1_ - d2_;
    if (fabsf(det) <= EPS) {
        auto sq = [](P2 a, P2 b) { float dx = a.x - b.x, dy = a.y - b.y; float qx = dx * dx, qy = dy * dy; return qx + qy; };
        float d1 = sq(p0, p1), d2 = sq(p0, p2), d3 = sq(p1, p2);
        float mx = d1 > d2 ? d1 : d2; mx = mx > d3 ? mx : d3;
        float h = sqrtf(mx) * 0.5f; radius = h + EPS;
        if (d1 >= d2 && d1 >= d3) c = P2{(p0.x + p1.x) * 0.5f, (p0.y + p1.y) * 0.5f};
        else if (d2 >= d1 && d2 >= d3) c = P2{(p0.x + p2.x) * 0.5f, (p0.y + p2.y) * 0.5f};
        else c = P2{(p1.x + p2.x) * 0.5f, (p1.y + p2.y) * 0.5f};
        return;
    }
    float n1 = c1 * v2.y, n2 = c2 * v1.y; float cx = (n1 - n2) / det;
    float n3 = v1.x * c2, n4 = v2.x * c1; float cy = (n3 - n4) / det;
    c.x = cx; c.y = cy;
    cx -= p0.x; cy -= p0.y;
    float qx = cx * cx, qy = cy * cy;
    radius = sqrtf(qx + qy) + EPS;
}
ORIP_HD inline void mec_third(const int32_t* xy, int i, int j, P2& c, float& radius) {
    const float EPS = 1.0e-4f;
    P2 pi = ipt(xy, i), pj = ipt(xy, j);
    c.x = (pj.x + pi.x) / 2.0f; c.y = (pj.y + pi.y) / 2.0f;
    radius = (float)nrm2(pj.x - pi.x, pj.y - pi.y) / 2.0f + EPS;
    for (int k = 0; k < j; ++k) {
        P2 pk = ipt(xy, k);
        if (nrm2(c.x - pk.x, c.y - pk.y) < (double)radius) continue;
        P2 nc{0, 0}; float nr = 0;
        mec_circle3(pi, pj, pk, nc, nr);
        if (nr > 0) { radius = nr; c = nc; }
    }
}
ORIP_HD inline void mec_second(const int32_t* xy, int i, P2& c, float& radius) {
    const float EPS = 1.0e-4f;
    P2 p0 = ipt(xy, 0), pi = ipt(xy, i);
    c.x = (p0.x + pi.x) / 2.0f; c.y = (p0.y + pi.y) / 2.0f;
    radius = (float)nrm2(p0.x - pi.x, p0.y - pi.y) / 2.0f + EPS;
    for (int j = 1; j < i; ++j) {
        P2 pj = ipt(xy, j);
        if (nrm2(c.x - pj.x, c.y - pj.y) < (double)radius) continue;
        P2 nc{0, 0}; float nr = 0;
        mec_third(xy, i, j, nc, nr);
        if (nr > 0) { radius = nr; c = nc; }
    }
}
ORIP_HD inline void min_enclosing_circle(const int32_t* xy, int64_t n, float& cx, float& cy, float& r) {
    const float EPS = 1.0e-4f;
    cx = cy = 0.f; r = 0.f;
    if (n == 0) return;
    if (n == 1) { cx = (float)xy[0]; cy = (float)xy[1]; r = EPS; return; }
    P2 p0 = ipt(xy, 0), p1 = ipt(xy, 1);
    if (n == 2) { cx = (p0.x + p1.x) / 2.0f; cy = (p0.y + p1.y) / 2.0f; r = (float)(nrm2(p0.x - p1.x, p0.y - p1.y) / 2.0) + EPS; return; }
    P2 c{(p0.x + p1.x) / 2.0f, (p0.y + p1.y) / 2.0f};
    float radius = (float)nrm2(p0.x - p1.x, p0.y - p1.y) / 2.0f + EPS;
    for (int i = 2; i < (int)n; ++i) {
        P2 pi = ipt(xy, i);
        float d = (float)nrm2(pi.x - c.x, pi.y - c.y);
        if (d < radius) continue;
        P2 nc{0, 0}; float nr = 0;
        mec_second(xy, i, nc, nr);
        if (nr > 0) { radius = nr; c = nc; }
    }
    cx = c.x; cy = c.y; r = radius;
}

// ---- exact capsule predicate: pixel p within r of segment ab (thick cv2.line / polylines / circle restated) ----
ORIP_HD inline bool in_capsule(long long px, long long py, long long ax, long long ay, long long bx, long long by, long long r2) {
    long long dx = bx - ax, dy = by - ay, qx = px - ax, qy = py - ay;
    long long L2 = dx * dx + dy * dy, q2 = qx * qx + qy * qy;
    if (L2 == 0) return q2 <= r2;
    long long t = qx * dx + qy * dy;
    if (t <= 0) return q2 <= r2;
    if (t >= L2) { long long ex = px - bx, ey = py - by; return ex * ex + ey * ey <= r2; }
    return q2 * L2 - t * t <= r2 * L2;
}

// float64 2-vector norm as numpy evaluates np.linalg.norm(np.array(a)-np.array(b)) in the reference's environment
// (OpenBLAS ddot contracts to fma(dy,dy,dx*dx); pinned against numpy 2.2.6 in the build container)
ORIP_HD inline double norm2_f64(double dx, double dy) { return sqrt(fma(dy, dy, dx * dx)); }

}  // namespace vs
