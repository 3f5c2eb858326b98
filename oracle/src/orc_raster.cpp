// oracle/src/orc_raster.cpp -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// CPU restatement of the raster half of the hot path: stage 02 (Lab k-means layers),
// stage 03 (open/close, Gaussian, Canny), stage 04 (Zhang-Suen thinning, CCL, centerline walk).
// OpenCV primitives are restated from SURVEY.md Appendix B ("recalled, unverified vs real OpenCV").
#include "orc_common.h"
#include <cstring>
#include <cfloat>
#include <queue>

namespace orc {

// ----------------------------------------------------------------------------------------------
// cv2.cvtColor(BGR2LAB) on 8-bit input (02_color_extract.py:35).  SURVEY App. B.1.
// ----------------------------------------------------------------------------------------------
static inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

void build_lab_tables(uint16_t gamma_tab[256], uint16_t cbrt_tab[3072], int coeffs[9]) {
    const int gamma_shift = 3, lab_shift = 12, lab_shift2 = lab_shift + gamma_shift;
    for (int i = 0; i < 256; i++) {
        double x = (double)i / 255.0;
        double g = x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4);
        long v = std::lrint(255.0 * (1 << gamma_shift) * g);
        gamma_tab[i] = (uint16_t)std::min<long>(std::max<long>(v, 0), 65535);
    }
    for (int i = 0; i < 3072; i++) {
        double x = (double)i / (255.0 * (1 << gamma_shift));
        double y = x < 0.008856 ? x * 7.787 + 0.13793103448275862 : std::cbrt(x);
        long v = std::lrint((double)(1 << lab_shift2) * y);
        cbrt_tab[i] = (uint16_t)std::min<long>(std::max<long>(v, 0), 65535);
    }
    static const double M[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160,
                                0.072169, 0.019334, 0.119193, 0.950227};
    static const double D65[3] = {0.950456, 1.0, 1.088754};
    // coeffs[i*3 + c] multiplies (R,G,B)[c]
    for (int i = 0; i < 3; i++)
        for (int c = 0; c < 3; c++)
            coeffs[i * 3 + c] = (int)std::lrint((double)(1 << lab_shift) * M[i * 3 + c] / D65[i]);
}

void bgr2lab(const u8* bgr, size_t n, u8* lab) {
    static uint16_t gt[256], ct[3072];
    static int C[9];
    static bool init = false;
    if (!init) { build_lab_tables(gt, ct, C); init = true; }
    const int Lscale = (116 * 255 + 50) / 100;
    const int Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    for (size_t i = 0; i < n; i++) {
        int B = gt[bgr[3 * i]], G = gt[bgr[3 * i + 1]], R = gt[bgr[3 * i + 2]];
        int fX = ct[descale(R * C[0] + G * C[1] + B * C[2], 12)];
        int fY = ct[descale(R * C[3] + G * C[4] + B * C[5], 12)];
        int fZ = ct[descale(R * C[6] + G * C[7] + B * C[8], 12)];
        int L = descale(Lscale * fY + Lshift, 15);
        int a = descale(500 * (fX - fY) + 128 * (1 << 15), 15);
        int b = descale(200 * (fY - fZ) + 128 * (1 << 15), 15);
        lab[3 * i] = (u8)std::min(std::max(L, 0), 255);
        lab[3 * i + 1] = (u8)std::min(std::max(a, 0), 255);
        lab[3 * i + 2] = (u8)std::min(std::max(b, 0), 255);
    }
}

// ----------------------------------------------------------------------------------------------
// cv2.kmeans(sample, K, None, (EPS|MAX_ITER, 40, 0.5), 3, KMEANS_PP_CENTERS)  (02:46-49).
// cv::RNG (MWC, state 0xffffffff), kmeans++ with 3 trials, Lloyd iterations.
// ----------------------------------------------------------------------------------------------
struct CvRNG {
    uint64_t state = 0xffffffffULL;
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    double next_double() {
        unsigned t = next();
        return (double)(((uint64_t)t << 32) | next()) * 5.4210108624275221700372640043497e-20;
    }
};

static inline float norm_l2sqr3(const float* a, const float* b) {
    float s = 0.f;
    for (int j = 0; j < 3; j++) { float t = a[j] - b[j]; s += t * t; }
    return s;
}

static void centers_pp(const float* data, int N, int K, CvRNG& rng, int trials, float* out) {
    std::vector<int> centers(K);
    std::vector<float> buf((size_t)N * 3);
    float* dist = buf.data(); float* tdist = dist + N; float* tdist2 = tdist + N;
    double sum0 = 0;
    centers[0] = (int)(rng.next() % (unsigned)N);
    for (int i = 0; i < N; i++) {
        dist[i] = norm_l2sqr3(data + 3 * i, data + 3 * centers[0]);
        sum0 += dist[i];
    }
    for (int k = 1; k < K; k++) {
        double bestSum = DBL_MAX; int bestCenter = -1;
        for (int j = 0; j < trials; j++) {
            double p = rng.next_double() * sum0;
            int ci = 0;
            for (; ci < N - 1; ci++) { p -= dist[ci]; if (p <= 0) break; }
            double s = 0;
            for (int i = 0; i < N; i++) {
                tdist2[i] = std::min(norm_l2sqr3(data + 3 * i, data + 3 * ci), dist[i]);
                s += tdist2[i];
            }
            if (s < bestSum) { bestSum = s; bestCenter = ci; std::swap(tdist, tdist2); }
        }
        centers[k] = bestCenter; sum0 = bestSum; std::swap(dist, tdist);
    }
    for (int k = 0; k < K; k++)
        for (int j = 0; j < 3; j++) out[3 * k + j] = data[3 * centers[k] + j];
}

double kmeans_pp(const float* data, int N, int K, int attempts, int max_iter, double eps, float* centers_out) {
    attempts = std::max(attempts, 1);
    double epsilon = std::max(eps, 0.0); epsilon *= epsilon;
    int maxCount = std::min(std::max(max_iter, 2), 100);
    if (K == 1) { attempts = 1; maxCount = 2; }
    CvRNG rng;
    std::vector<float> centers(K * 3, 0.f), old_centers(K * 3, 0.f);
    std::vector<int> labels(N, 0), counters(K);
    std::vector<double> dists(N);
    double best = DBL_MAX;
    for (int a = 0; a < attempts; a++) {
        double compactness = 0;
        for (int iter = 0;;) {
            double max_center_shift = iter == 0 ? DBL_MAX : 0.0;
            std::swap(centers, old_centers);
            if (iter == 0) {
                centers_pp(data, N, K, rng, 3, centers.data());
            } else {
                std::fill(centers.begin(), centers.end(), 0.f);
                std::fill(counters.begin(), counters.end(), 0);
                for (int i = 0; i < N; i++) {
                    int k = labels[i];
                    for (int j = 0; j < 3; j++) centers[3 * k + j] += data[3 * i + j];
                    counters[k]++;
                }
                for (int k = 0; k < K; k++) {
                    if (counters[k] != 0) continue;
                    int max_k = 0;
                    for (int k1 = 1; k1 < K; k1++) if (counters[max_k] < counters[k1]) max_k = k1;
                    double max_dist = 0; int farthest_i = -1;
                    float* base = &centers[3 * max_k];
                    float nb[3]; float scale = 1.f / counters[max_k];
                    for (int j = 0; j < 3; j++) nb[j] = base[j] * scale;
                    for (int i = 0; i < N; i++) {
                        if (labels[i] != max_k) continue;
                        double d = norm_l2sqr3(data + 3 * i, nb);
                        if (max_dist <= d) { max_dist = d; farthest_i = i; }
                    }
                    counters[max_k]--; counters[k]++; labels[farthest_i] = k;
                    for (int j = 0; j < 3; j++) {
                        base[j] -= data[3 * farthest_i + j];
                        centers[3 * k + j] += data[3 * farthest_i + j];
                    }
                }
                for (int k = 0; k < K; k++) {
                    float scale = 1.f / counters[k];
                    for (int j = 0; j < 3; j++) centers[3 * k + j] *= scale;
                    if (iter > 0) {
                        double dist = 0;
                        for (int j = 0; j < 3; j++) {
                            double t = centers[3 * k + j] - old_centers[3 * k + j];
                            dist += t * t;
                        }
                        max_center_shift = std::max(max_center_shift, dist);
                    }
                }
            }
            bool last = (++iter == std::max(maxCount, 2) || max_center_shift <= epsilon);
            if (last) {
                compactness = 0;
                for (int i = 0; i < N; i++) {
                    dists[i] = norm_l2sqr3(data + 3 * i, &centers[3 * labels[i]]);
                    compactness += dists[i];
                }
                break;
            } else {
                for (int i = 0; i < N; i++) {
                    double md = DBL_MAX; int kb = 0;
                    for (int k = 0; k < K; k++) {
                        double d = norm_l2sqr3(data + 3 * i, &centers[3 * k]);
                        if (md > d) { md = d; kb = k; }
                    }
                    labels[i] = kb;
                }
            }
        }
        if (compactness < best) {
            best = compactness;
            std::copy(centers.begin(), centers.end(), centers_out);
        }
    }
    return best;
}

// numpy assignment of all pixels (02:53-55): float32 diffs, (d0^2 + d1^2) + d2^2, first minimum.
void assign_labels(const u8* lab, size_t n, const float* centers, int K, int32_t* labels) {
    for (size_t i = 0; i < n; i++) {
        float p0 = lab[3 * i], p1 = lab[3 * i + 1], p2 = lab[3 * i + 2];
        float best = 0; int kb = 0;
        for (int k = 0; k < K; k++) {
            float d0 = p0 - centers[3 * k], d1 = p1 - centers[3 * k + 1], d2 = p2 - centers[3 * k + 2];
            float q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2;
            float s = q0 + q1; s = s + q2;
            if (k == 0 || s < best) { best = s; kb = k; }
        }
        labels[i] = kb;
    }
}

// ----------------------------------------------------------------------------------------------
// cv2.getStructuringElement + erode/dilate (02:136-154, 03:23-30).  SURVEY App. B.3.
// ----------------------------------------------------------------------------------------------
void make_se(int shape, int k, std::vector<u8>& se) {
    se.assign((size_t)k * k, 0);
    if (shape == 0) { std::fill(se.begin(), se.end(), 1); return; }
    // MORPH_ELLIPSE
    int r = k / 2, c = k / 2;
    double inv_r2 = r ? 1.0 / ((double)r * r) : 0;
    for (int i = 0; i < k; i++) {
        int dy = i - r, j1 = 0, j2 = 0;
        if (std::abs(dy) <= r) {
            int dx = (int)std::lrint(c * std::sqrt((r * r - dy * dy) * inv_r2));
            j1 = std::max(c - dx, 0); j2 = std::min(c + dx + 1, k);
        }
        for (int j = j1; j < j2; j++) se[(size_t)i * k + j] = 1;
    }
}

void morph(const u8* src, u8* dst, int H, int W, const u8* se, int kh, int kw, bool dilate) {
    int ay = kh / 2, ax = kw / 2;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int v = dilate ? 0 : 255;
            for (int i = 0; i < kh; i++) {
                int yy = y + i - ay; if (yy < 0 || yy >= H) continue;
                for (int j = 0; j < kw; j++) {
                    if (!se[i * kw + j]) continue;
                    int xx = x + j - ax; if (xx < 0 || xx >= W) continue;
                    int s = src[(size_t)yy * W + xx];
                    v = dilate ? std::max(v, s) : std::min(v, s);
                }
            }
            dst[(size_t)y * W + x] = (u8)v;
        }
}

void morph_open_close(u8* img, int H, int W, int shape, int k, int open_iters, int close_iters) {
    std::vector<u8> se; make_se(shape, k, se);
    std::vector<u8> tmp((size_t)H * W);
    auto pass = [&](bool dil) { morph(img, tmp.data(), H, W, se.data(), k, k, dil); memcpy(img, tmp.data(), tmp.size()); };
    if (open_iters > 0) {
        for (int i = 0; i < open_iters; i++) pass(false);
        for (int i = 0; i < open_iters; i++) pass(true);
    }
    if (close_iters > 0) {
        for (int i = 0; i < close_iters; i++) pass(true);
        for (int i = 0; i < close_iters; i++) pass(false);
    }
}

// ----------------------------------------------------------------------------------------------
// cv2.GaussianBlur(u8, (k,k), 0) (03:33).  SURVEY App. B.4: fixed tables, exact products,
// one round-half-up at the end, BORDER_REFLECT_101.
// ----------------------------------------------------------------------------------------------
static inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * (n - 1) - p; }
    return p;
}

int gaussian_blur(const u8* src, u8* dst, int H, int W, int k) {
    static const int w3[3] = {1, 2, 1}, w5[5] = {1, 4, 6, 4, 1}, w7[7] = {8, 28, 56, 72, 56, 28, 8};
    const int* w; int shift;
    if (k == 3) { w = w3; shift = 4; } else if (k == 5) { w = w5; shift = 8; } else if (k == 7) { w = w7; shift = 16; }
    else return -1;
    int r = k / 2;
    std::vector<int> hrow((size_t)H * W);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int s = 0;
            for (int j = 0; j < k; j++) s += w[j] * src[(size_t)y * W + reflect101(x + j - r, W)];
            hrow[(size_t)y * W + x] = s;
        }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int64_t s = 0;
            for (int i = 0; i < k; i++) s += (int64_t)w[i] * hrow[(size_t)reflect101(y + i - r, H) * W + x];
            dst[(size_t)y * W + x] = (u8)((s + ((int64_t)1 << (shift - 1))) >> shift);
        }
    return 0;
}

// ----------------------------------------------------------------------------------------------
// cv2.Canny(u8, low, high), aperture 3, L1 gradient (03:34).  SURVEY App. B.5.
// ----------------------------------------------------------------------------------------------
void canny(const u8* src, u8* dst, int H, int W, int low, int high) {
    if (low > high) std::swap(low, high);
    auto at = [&](int y, int x) -> int {
        y = std::min(std::max(y, 0), H - 1); x = std::min(std::max(x, 0), W - 1);
        return src[(size_t)y * W + x];
    };
    std::vector<int16_t> dx((size_t)H * W), dy((size_t)H * W);
    std::vector<int> mag((size_t)(H + 2) * (W + 2), 0);
    const int MW = W + 2;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int gx = (at(y - 1, x + 1) + 2 * at(y, x + 1) + at(y + 1, x + 1)) -
                     (at(y - 1, x - 1) + 2 * at(y, x - 1) + at(y + 1, x - 1));
            int gy = (at(y + 1, x - 1) + 2 * at(y + 1, x) + at(y + 1, x + 1)) -
                     (at(y - 1, x - 1) + 2 * at(y - 1, x) + at(y - 1, x + 1));
            dx[(size_t)y * W + x] = (int16_t)gx; dy[(size_t)y * W + x] = (int16_t)gy;
            mag[(size_t)(y + 1) * MW + x + 1] = std::abs(gx) + std::abs(gy);
        }
    // map: 0 weak candidate, 1 not an edge, 2 edge; 1-px border of 1
    std::vector<u8> map((size_t)(H + 2) * MW, 1);
    std::vector<size_t> stack;
    const int TG22 = 13573;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int* m_a = &mag[(size_t)(y + 1) * MW + x + 1];
            const int* m_p = m_a - MW; const int* m_n = m_a + MW;
            int m = *m_a; u8 res = 1;
            if (m > low) {
                int xs = dx[(size_t)y * W + x], ys = dy[(size_t)y * W + x];
                int ax = std::abs(xs); int ay = std::abs(ys) << 15;
                int tg22x = ax * TG22; bool keep = false;
                if (ay < tg22x) { keep = (m > m_a[-1] && m >= m_a[1]); }
                else {
                    int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) keep = (m > m_p[0] && m >= m_n[0]);
                    else { int s = (xs ^ ys) < 0 ? -1 : 1; keep = (m > m_p[-s] && m > m_n[s]); }
                }
                if (keep) res = (m > high) ? 2 : 0;
            }
            size_t idx = (size_t)(y + 1) * MW + x + 1;
            map[idx] = res;
            if (res == 2) stack.push_back(idx);
        }
    while (!stack.empty()) {
        size_t i = stack.back(); stack.pop_back();
        const long nb[8] = {-MW - 1, -MW, -MW + 1, -1, 1, MW - 1, MW, MW + 1};
        for (long d : nb) { size_t j = (size_t)((long)i + d); if (map[j] == 0) { map[j] = 2; stack.push_back(j); } }
    }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) dst[(size_t)y * W + x] = map[(size_t)(y + 1) * MW + x + 1] == 2 ? 255 : 0;
}

// ----------------------------------------------------------------------------------------------
// Zhang-Suen thinning.  Neighbour numbering is a parameter: 04:53-55 derives a 180-degree rotated
// numbering from its _shift() arguments, 08:349-352 uses the standard one.
// offs[i] = (dy,dx) of P2..P9.
// ----------------------------------------------------------------------------------------------
static int zs_generic(u8* img /*0/1*/, int H, int W, const int offs[8][2], int max_iter) {
    // examines only the current foreground pixels (kept as a list in raster order); same deletions as a full-image pass
    std::vector<uint32_t> act, nxt; std::vector<size_t> del;
    for (size_t i = 0; i < (size_t)H * W; i++) if (img[i]) act.push_back((uint32_t)i);
    auto get = [&](int y, int x) -> int { return (y < 0 || y >= H || x < 0 || x >= W) ? 0 : img[(size_t)y * W + x]; };
    int it = 0; bool changed = true;
    while (changed && it < max_iter) {
        it++; changed = false;
        for (int sub = 0; sub < 2; sub++) {
            del.clear(); nxt.clear();
            for (uint32_t idx : act) {
                int y = (int)(idx / W), x = (int)(idx % W);
                int P[8];
                for (int i = 0; i < 8; i++) P[i] = get(y + offs[i][0], x + offs[i][1]);
                int B = 0; for (int i = 0; i < 8; i++) B += P[i];
                bool kill = false;
                if (B >= 2 && B <= 6) {
                    int A = 0; for (int i = 0; i < 8; i++) A += (P[i] == 0 && P[(i + 1) & 7] == 1);
                    if (A == 1) {
                        int P2 = P[0], P4 = P[2], P6 = P[4], P8 = P[6];
                        kill = sub == 0 ? (P2 * P4 * P6 == 0 && P4 * P6 * P8 == 0) : (P2 * P4 * P8 == 0 && P2 * P6 * P8 == 0);
                    }
                }
                if (kill) del.push_back(idx); else nxt.push_back(idx);
            }
            if (!del.empty()) { changed = true; for (size_t i : del) img[i] = 0; }
            act.swap(nxt);
        }
    }
    return it;
}

int thinning_rot(const u8* edges, u8* skel, int H, int W) {  // 04:35-99
    static const int offs[8][2] = {{1, 0}, {1, -1}, {0, -1}, {-1, -1}, {-1, 0}, {-1, 1}, {0, 1}, {1, 1}};
    std::vector<u8> b((size_t)H * W);
    bool any = false;
    for (size_t i = 0; i < b.size(); i++) { b[i] = edges[i] > 0; any |= b[i]; }
    int it = 0;
    if (any) it = zs_generic(b.data(), H, W, offs, 120);
    for (size_t i = 0; i < b.size(); i++) skel[i] = b[i] ? 255 : 0;
    return it;
}

int zhang_suen_std(const u8* src, u8* dst, int H, int W, int max_iter) {  // 08:342-372
    static const int offs[8][2] = {{-1, 0}, {-1, 1}, {0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, -1}};
    std::vector<u8> b((size_t)H * W);
    for (size_t i = 0; i < b.size(); i++) b[i] = src[i] > 0;
    int it = zs_generic(b.data(), H, W, offs, max_iter);
    for (size_t i = 0; i < b.size(); i++) dst[i] = b[i] ? 255 : 0;
    return it;
}

// ----------------------------------------------------------------------------------------------
// cv2.connectedComponents(connectivity=8) (04:111, 08:421).  SURVEY App. B.6: label numbering is
// algorithm dependent; restated as the order block-based (2x2) scanners produce: components are
// numbered by the raster position of the first 2x2 block that contains one of their pixels.
// ----------------------------------------------------------------------------------------------
int ccl8(const u8* fg, int32_t* labels, int H, int W) {
    size_t N = (size_t)H * W;
    std::fill(labels, labels + N, 0);
    std::vector<size_t> stack;
    std::vector<int64_t> key;  // per provisional component
    int Wb = (W + 1) / 2;
    int n = 0;
    for (size_t s = 0; s < N; s++) {
        if (!fg[s] || labels[s]) continue;
        n++; labels[s] = n; stack.push_back(s);
        int64_t k = INT64_MAX;
        while (!stack.empty()) {
            size_t i = stack.back(); stack.pop_back();
            int y = (int)(i / W), x = (int)(i % W);
            k = std::min(k, (int64_t)(y >> 1) * Wb + (x >> 1));
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    size_t j = (size_t)yy * W + xx;
                    if (fg[j] && !labels[j]) { labels[j] = n; stack.push_back(j); }
                }
        }
        key.push_back(k);
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
    std::vector<int32_t> remap(n + 1, 0);
    for (int r = 0; r < n; r++) remap[order[r] + 1] = r + 1;
    for (size_t i = 0; i < N; i++) labels[i] = remap[labels[i]];
    return n;
}

// ----------------------------------------------------------------------------------------------
// trace_centerlines (04:102-211).
// ----------------------------------------------------------------------------------------------
void trace_centerlines(const u8* skel, int H, int W, PolyList& out) {
    static const int NB[8][2] = {{-1, -1}, {0, -1}, {1, -1}, {-1, 0}, {1, 0}, {-1, 1}, {0, 1}, {1, 1}};  // (dx,dy) 04:12
    size_t N = (size_t)H * W;
    std::vector<u8> S(N);
    int64_t total_fg = 0;
    for (size_t i = 0; i < N; i++) { S[i] = skel[i] > 0; total_fg += S[i]; }
    if (!total_fg) return;
    std::vector<int32_t> lab(N);
    int num = ccl8(S.data(), lab.data(), H, W);
    // per-component pixel lists in raster order
    std::vector<int64_t> cnt(num + 2, 0);
    for (size_t i = 0; i < N; i++) if (lab[i]) cnt[lab[i] + 1]++;
    for (int c = 1; c <= num + 1; c++) cnt[c] += cnt[c - 1];
    std::vector<uint32_t> pix((size_t)total_fg);
    {
        std::vector<int64_t> pos(cnt.begin(), cnt.end());
        for (size_t i = 0; i < N; i++) if (lab[i]) pix[(size_t)pos[lab[i]]++] = (uint32_t)i;
    }
    // degree = number of 8-neighbours in the same component (== skeleton neighbours)
    std::vector<u8> deg(N, 0), visited(N, 0);
    for (size_t i = 0; i < N; i++) {
        if (!S[i]) continue;
        int y = (int)(i / W), x = (int)(i % W), d = 0;
        for (int k = 0; k < 8; k++) {
            int xx = x + NB[k][0], yy = y + NB[k][1];
            if (xx >= 0 && xx < W && yy >= 0 && yy < H && S[(size_t)yy * W + xx]) d++;
        }
        deg[i] = (u8)d;
    }
    for (int c = 1; c <= num; c++) {
        int64_t b = cnt[c], e = cnt[c + 1], fg_comp = e - b;
        // phase 1: from endpoints (04:144-171)
        for (int64_t q = b; q < e; q++) {
            size_t s = pix[(size_t)q];
            if (deg[s] != 1 || visited[s]) continue;
            int x0 = (int)(s % W), y0 = (int)(s / W);
            out.push_pt(x0, y0); visited[s] = 1;
            int px = x0, py = y0, pvx = -1, pvy = -1; bool has_prev = false;
            int64_t guard = 0;
            while (true) {
                int nx = -1, ny = -1;
                for (int k = 0; k < 8; k++) {
                    int xx = px + NB[k][0], yy = py + NB[k][1];
                    if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                    size_t j = (size_t)yy * W + xx;
                    if (!S[j]) continue;
                    if (has_prev && xx == pvx && yy == pvy) continue;
                    if (visited[j]) continue;
                    nx = xx; ny = yy; break;
                }
                if (nx < 0) break;
                out.push_pt(nx, ny); visited[(size_t)ny * W + nx] = 1;
                pvx = px; pvy = py; has_prev = true; px = nx; py = ny;
                u8 d = deg[(size_t)py * W + px];
                if (d >= 3 || d == 1) break;
                guard++;
                if (guard > total_fg * 2) break;
            }
            if (out.open_len() >= 2) out.end_poly(); else out.abort_poly();
        }
        // phase 2: leftovers / cycles (04:174-205)
        for (int64_t q = b; q < e; q++) {
            size_t s = pix[(size_t)q];
            if (visited[s]) continue;
            int x0 = (int)(s % W), y0 = (int)(s / W);
            out.push_pt(x0, y0); visited[s] = 1;
            int px = x0, py = y0, pvx = -1, pvy = -1; bool has_prev = false;
            int64_t guard = 0;
            while (true) {
                int nx = -1, ny = -1, ax = -1, ay = -1;
                for (int k = 0; k < 8; k++) {
                    int xx = px + NB[k][0], yy = py + NB[k][1];
                    if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                    size_t j = (size_t)yy * W + xx;
                    if (!S[j]) continue;
                    if (has_prev && xx == pvx && yy == pvy) continue;
                    if (ax < 0) { ax = xx; ay = yy; }
                    if (!visited[j]) { nx = xx; ny = yy; break; }
                }
                if (nx < 0) { nx = ax; ny = ay; }
                if (nx < 0) break;
                out.push_pt(nx, ny); visited[(size_t)ny * W + nx] = 1;
                pvx = px; pvy = py; has_prev = true; px = nx; py = ny;
                if (px == x0 && py == y0) break;
                guard++;
                if (guard > fg_comp * 4) break;
            }
            size_t n = out.open_len();
            if (n >= 2) {
                const int32_t* p = out.pts.data() + 2 * out.off.back();
                int32_t fx = p[0], fy = p[1];
                double hx = (double)fx - p[2 * (n - 1)], hy = (double)fy - p[2 * (n - 1) + 1];
                if (std::hypot(hx, hy) < 1.5) out.push_pt(fx, fy);
                out.end_poly();
            } else out.abort_poly();
        }
    }
}


// ------------------------------------------------------------------------------------------------
// 01_resize.py:19 -- cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_AREA), shrinking only.
// PARITY UNPINNED (cv2 is absent): restated from OpenCV 4.x imgproc/src/resize.cpp as published --
//   * both ratios integers (|scale - round| < DBL_EPSILON): resizeAreaFast_: 2 x 2 -> (a + b + c + d + 2) >> 2; otherwise the integer cell
//     sum times float(1 / area), rounded half-to-even (saturate_cast<uchar>(float) is cvRound);
//   * otherwise resizeArea_: per axis a table of (source index, destination index, float alpha) from computeResizeAreaTab (partial cell on the left
//     when more than 1e-3 of it is inside, whole cells with alpha 1 / cellWidth, partial cell on the right), rows reduced in float in table order
//     (buf += S * alpha), then destination rows in float (sum = beta * buf for the first source row of a destination row, sum += beta * buf after).
// The x86 wheels run this code path without fused multiply-add (baseline SSE build), which is what -ffp-contract=off gives here.
// ------------------------------------------------------------------------------------------------
namespace {
struct AreaTab { int si, di; float alpha; };
void area_tab(int ssize, int dsize, double scale, std::vector<AreaTab>& tab) {
    tab.clear();
    for (int dx = 0; dx < dsize; dx++) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) tab.push_back({sx1 - 1, dx, (float)((sx1 - fsx1) / cell)});
        for (int sx = sx1; sx < sx2; sx++) tab.push_back({sx, dx, (float)(1.0 / cell)});
        if (fsx2 - sx2 > 1e-3) tab.push_back({sx2, dx, (float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell)});
    }
}
inline u8 round_u8(float v) { const long r = lrintf(v); return (u8)(r < 0 ? 0 : r > 255 ? 255 : r); }   // default rounding mode: half to even
}  // namespace

int resize_area(const u8* src, int sh, int sw, int cn, u8* dst, int dh, int dw) {
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || dh > sh || dw > sw || cn < 1 || cn > 4) return -1;
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    const int isx = (int)lrint(scale_x), isy = (int)lrint(scale_y);
    if (std::abs(scale_x - isx) < DBL_EPSILON && std::abs(scale_y - isy) < DBL_EPSILON) {
        const int area = isx * isy; const float inv = 1.f / area;
        for (int dy = 0; dy < dh; dy++)
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    int sum = 0;
                    for (int sy = 0; sy < isy; sy++)
                        for (int sx = 0; sx < isx; sx++) sum += src[((size_t)(dy * isy + sy) * sw + dx * isx + sx) * cn + c];
                    dst[((size_t)dy * dw + dx) * cn + c] = (isx == 2 && isy == 2) ? (u8)((sum + 2) >> 2) : round_u8(sum * inv);
                }
        return 0;
    }
    std::vector<AreaTab> xt, yt;
    area_tab(sw, dw, scale_x, xt); area_tab(sh, dh, scale_y, yt);
    std::vector<float> buf((size_t)dw * cn), sum((size_t)dw * cn, 0.f);
    int prev_dy = yt.empty() ? 0 : yt[0].di;
    for (size_t j = 0; j < yt.size(); j++) {
        const float beta = yt[j].alpha; const int dy = yt[j].di;
        const u8* S = src + (size_t)yt[j].si * sw * cn;
        std::fill(buf.begin(), buf.end(), 0.f);
        for (const AreaTab& e : xt)
            for (int c = 0; c < cn; c++) buf[(size_t)e.di * cn + c] = buf[(size_t)e.di * cn + c] + S[(size_t)e.si * cn + c] * e.alpha;
        if (dy != prev_dy) {
            u8* D = dst + (size_t)prev_dy * dw * cn;
            for (size_t i = 0; i < sum.size(); i++) { D[i] = round_u8(sum[i]); sum[i] = beta * buf[i]; }
            prev_dy = dy;
        } else {
            for (size_t i = 0; i < sum.size(); i++) sum[i] += beta * buf[i];
        }
    }
    u8* D = dst + (size_t)prev_dy * dw * cn;
    for (size_t i = 0; i < sum.size(); i++) D[i] = round_u8(sum[i]);
    return 0;
}
}  // namespace orc
