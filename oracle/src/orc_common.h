// oracle/src/orc_common.h
// TEST INFRASTRUCTURE ONLY.  CPU restatement ("oracle") of the hot path of
// omnirevolve-image-processor (stages 02 -> 12).  Nothing under oracle/ is part of the
// product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// All file:line citations refer to /root/reference/image_processor/.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cmath>
#include <vector>
#include <algorithm>

namespace orc {

typedef uint8_t u8;

// A list of integer polylines, flattened: polyline i = pts[2*off[i] .. 2*off[i+1]) as (x,y) pairs.
struct PolyList {
    std::vector<int64_t> off{0};
    std::vector<int32_t> pts;
    size_t count() const { return off.size() - 1; }
    size_t npts(size_t i) const { return (size_t)(off[i + 1] - off[i]); }
    const int32_t* p(size_t i) const { return pts.data() + 2 * off[i]; }
    void begin() {}
    void push_pt(int32_t x, int32_t y) { pts.push_back(x); pts.push_back(y); }
    void end_poly() { off.push_back((int64_t)(pts.size() / 2)); }
    void add(const int32_t* xy, size_t n) {
        pts.insert(pts.end(), xy, xy + 2 * n);
        end_poly();
    }
    // drop points pushed since the last end_poly()
    void abort_poly() { pts.resize((size_t)off.back() * 2); }
    size_t open_len() const { return pts.size() / 2 - (size_t)off.back(); }
    void clear() { off.assign(1, 0); pts.clear(); }
};

struct Taps {
    std::vector<int32_t> xy;  // (x,y) pairs
    size_t count() const { return xy.size() / 2; }
    void add(int32_t x, int32_t y) { xy.push_back(x); xy.push_back(y); }
};

// Python round() / numpy rint: round-half-to-even.
static inline int64_t round_half_even(double v) { return (int64_t)std::nearbyint(v); }

// numpy's float32 pairwise summation (umath/loops_utils.h @TYPE@_pairwise_sum), which is what
// ndarray.sum() of a contiguous 1-D float32 array evaluates (checked against numpy 2.2.6).
static inline float np_pairwise_sum_f32(const float* a, size_t n) {
    if (n < 8) {
        float r = 0.f;
        for (size_t i = 0; i < n; i++) r += a[i];
        return r;
    } else if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        size_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        size_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum_f32(a, n2) + np_pairwise_sum_f32(a + n2, n - n2);
    }
}

// float32 hypot, correctly rounded (glibc >= 2.35 hypotf, which numpy's np.hypot(float32) calls).
static inline float hypot_f32(float dx, float dy) {
    return (float)std::sqrt((double)dx * (double)dx + (double)dy * (double)dy);
}

// ---- raster (orc_raster.cpp) ----
void build_lab_tables(uint16_t gamma_tab[256], uint16_t cbrt_tab[3072], int coeffs[9]);
void bgr2lab(const u8* bgr, size_t n, u8* lab);
double kmeans_pp(const float* data, int N, int K, int attempts, int max_iter, double eps,
                 float* centers_out);
void assign_labels(const u8* lab, size_t n, const float* centers, int K, int32_t* labels);
void morph(const u8* src, u8* dst, int H, int W, const u8* se, int kh, int kw, bool dilate);
void make_se(int shape /*0 rect, 2 ellipse*/, int k, std::vector<u8>& se);
void morph_open_close(u8* img, int H, int W, int shape, int k, int open_iters, int close_iters);
int gaussian_blur(const u8* src, u8* dst, int H, int W, int k);
int resize_area(const u8* src, int sh, int sw, int cn, u8* dst, int dh, int dw);
void canny(const u8* src, u8* dst, int H, int W, int low, int high);
int thinning_rot(const u8* edges, u8* skel, int H, int W);          // 04:35-99
int zhang_suen_std(const u8* src, u8* dst, int H, int W, int max_iter);  // 08:342-372
int ccl8(const u8* fg, int32_t* labels, int H, int W);              // block-raster label order
void trace_centerlines(const u8* skel, int H, int W, PolyList& out);    // 04:102-211

// ---- vector (orc_vector.cpp) ----
struct Params08 {
    double tap_diam = 60, tap_max_dim = 25, min_keep = 12, tap_max_per = 160;
    int tap_max_v = 50;
    double sample_step = 8, tail_len_px = 120, col_rad = 18, grid_stride = 18, max_jump = 80;
    int post_on = 1, post_brush = 16;
    double post_step = 6, post_eps = 1.28;
    int post_minlen = 32;
    int W = 8400, H = 11880;
    int brush_forbid = 36;
};
struct Params10 {
    double tap_diam = 60, min_keep = 12, tap_max_per = 150;
    int tap_max_v = 50;
    double max_jump = 80, D_lines = 120, D_taps = 120, step_px = 1.0;
    int W = 8400, H = 11880;
};

void stamp_capsule(u8* mask, int H, int W, int x0, int y0, int x1, int y1, int r, u8 val);
void stamp_disc(u8* mask, int H, int W, int cx, int cy, int r, u8 val);
double arc_length_i32(const int32_t* xy, size_t n, bool closed);
float poly_perimeter_f32(const int32_t* xy, size_t n);
void min_enclosing_circle_f32(const float* xy, size_t n, float& cx, float& cy, float& r);
void scale_polys(const PolyList& in, float sx, float sy, float dx, float dy, PolyList& out);  // 05:82-96
void sort_contours07(const PolyList& in, PolyList& out);                                      // 07:19-95
void resample_arclen(const float* xy, size_t n, bool closed_in, double step,
                     std::vector<double>& out, bool& is_f32_passthrough);                     // 08:53-64
void split_on_long_jumps08(const int32_t* xy, size_t n, double max_jump, PolyList& out);      // 08:185-196
void split_on_long_jumps10(const int32_t* xy, size_t n, double max_jump, PolyList& out);      // 10:49-63
void split_small_and_taps08(const PolyList& in, const Params08& P, PolyList& kept, Taps& taps);  // 08:198-216
void virtual_draw08(const int32_t* xy, size_t n, const Params08& P, u8* mask, PolyList& segs);   // 08:117-183
void reorder_only(const PolyList& in, PolyList& out, int length_kind /*0: 08 perimeter f32, 1: 10 arcLength*/);
void cluster_by_overlap(const std::vector<int32_t>& bboxes, std::vector<std::vector<int>>& groups);  // 08:319-338
void bfs_path(const u8* img, int h, int w, int sy, int sx, int gy, int gx, std::vector<int32_t>& path_yx);  // 08:261-280
void component_best_path(const u8* comp, int h, int w, bool has_a, int ay, int ax, bool has_b, int by,
                         int bx, int min_len, std::vector<int32_t>& path_yx);                 // 08:295-317
void post_skeleton_merge(const PolyList& lines, const Params08& P, PolyList& out);            // 08:376-469
void stage08_layer(const PolyList& sorted, const Params08& P, PolyList& lines, Taps& taps);   // 08:484-557
void cut_poly_against_mask(const int32_t* xy, size_t n, const u8* forb, int H, int W, double step_px,
                           PolyList& out);                                                    // 10:142-177
void tiny_and_taps10(const PolyList& in, const Params10& P, PolyList& kept, Taps& taps);      // 10:99-118
void stage10_layer(const PolyList& lines_in, const Taps& taps_in, const Params10& P, u8* forbidden,
                   PolyList& lines_out, Taps& taps_out);                                      // 10:236-267

struct Op { int type; /*0 line, 1 tap*/ int line_idx; int flip; int x, y; };
void build_ops12(const PolyList& lines, const Taps& taps, double R_insert, std::vector<Op>& ops);  // 12:85-187

}  // namespace orc
