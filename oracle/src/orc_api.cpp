// oracle/src/orc_api.cpp -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// Flat C ABI over the CPU restatement, loaded with ctypes by oracle/oracle.py.
#include "orc_common.h"
#include <cstring>

using namespace orc;

extern "C" {

// ---- handles for variable-length results ----
void* orc_pl_new() { return new PolyList(); }
void orc_pl_free(void* h) { delete (PolyList*)h; }
int64_t orc_pl_count(void* h) { return (int64_t)((PolyList*)h)->count(); }
int64_t orc_pl_total(void* h) { return (int64_t)(((PolyList*)h)->pts.size() / 2); }
void orc_pl_get(void* h, int64_t* off, int32_t* pts) {
    PolyList* p = (PolyList*)h;
    memcpy(off, p->off.data(), p->off.size() * sizeof(int64_t));
    if (!p->pts.empty()) memcpy(pts, p->pts.data(), p->pts.size() * sizeof(int32_t));
}
void orc_pl_set(void* h, int64_t n, const int64_t* off, const int32_t* pts) {
    PolyList* p = (PolyList*)h;
    p->off.assign(off, off + n + 1);
    p->pts.assign(pts, pts + 2 * off[n]);
}
void* orc_taps_new() { return new Taps(); }
void orc_taps_free(void* h) { delete (Taps*)h; }
int64_t orc_taps_count(void* h) { return (int64_t)((Taps*)h)->count(); }
void orc_taps_get(void* h, int32_t* xy) { Taps* t = (Taps*)h; if (!t->xy.empty()) memcpy(xy, t->xy.data(), t->xy.size() * 4); }
void orc_taps_set(void* h, int64_t n, const int32_t* xy) { ((Taps*)h)->xy.assign(xy, xy + 2 * n); }

// ---- raster ----
void orc_lab_tables(uint16_t* gamma_tab, uint16_t* cbrt_tab, int* coeffs) { build_lab_tables(gamma_tab, cbrt_tab, coeffs); }
void orc_bgr2lab(const u8* bgr, int64_t n, u8* lab) { bgr2lab(bgr, (size_t)n, lab); }
double orc_kmeans(const float* data, int N, int K, int attempts, int max_iter, double eps, float* centers) {
    return kmeans_pp(data, N, K, attempts, max_iter, eps, centers);
}
void orc_assign(const u8* lab, int64_t n, const float* centers, int K, int32_t* labels) { assign_labels(lab, (size_t)n, centers, K, labels); }
void orc_morph_open_close(u8* img, int H, int W, int shape, int k, int open_iters, int close_iters) {
    morph_open_close(img, H, W, shape, k, open_iters, close_iters);
}
void orc_make_se(int shape, int k, u8* se) { std::vector<u8> v; make_se(shape, k, v); memcpy(se, v.data(), v.size()); }
int orc_resize_area(const u8* src, int sh, int sw, int cn, u8* dst, int dh, int dw) { return resize_area(src, sh, sw, cn, dst, dh, dw); }
int orc_gaussian(const u8* src, u8* dst, int H, int W, int k) { return gaussian_blur(src, dst, H, W, k); }
void orc_canny(const u8* src, u8* dst, int H, int W, int low, int high) { canny(src, dst, H, W, low, high); }
int orc_thin_rot(const u8* e, u8* s, int H, int W) { return thinning_rot(e, s, H, W); }
int orc_zs_std(const u8* e, u8* s, int H, int W, int max_iter) { return zhang_suen_std(e, s, H, W, max_iter); }
int orc_ccl8(const u8* fg, int32_t* labels, int H, int W) { return ccl8(fg, labels, H, W); }
void orc_trace(const u8* skel, int H, int W, void* out) { ((PolyList*)out)->clear(); trace_centerlines(skel, H, W, *(PolyList*)out); }

// ---- vector ----
static Params08 p08(const double* a) {
    Params08 P;
    if (!a) return P;
    P.tap_diam = a[0]; P.tap_max_dim = a[1]; P.min_keep = a[2]; P.tap_max_per = a[3]; P.tap_max_v = (int)a[4];
    P.sample_step = a[5]; P.tail_len_px = a[6]; P.col_rad = a[7]; P.grid_stride = a[8]; P.max_jump = a[9];
    P.post_on = (int)a[10]; P.post_brush = (int)a[11]; P.post_step = a[12]; P.post_eps = a[13]; P.post_minlen = (int)a[14];
    P.W = (int)a[15]; P.H = (int)a[16]; P.brush_forbid = (int)a[17];
    return P;
}
static Params10 p10(const double* a) {
    Params10 P;
    if (!a) return P;
    P.tap_diam = a[0]; P.min_keep = a[1]; P.tap_max_per = a[2]; P.tap_max_v = (int)a[3]; P.max_jump = a[4];
    P.D_lines = a[5]; P.D_taps = a[6]; P.step_px = a[7]; P.W = (int)a[8]; P.H = (int)a[9];
    return P;
}

void orc_stamp_capsule(u8* mask, int H, int W, int x0, int y0, int x1, int y1, int r) { stamp_capsule(mask, H, W, x0, y0, x1, y1, r, 255); }
double orc_arc_length(const int32_t* xy, int64_t n, int closed) { return arc_length_i32(xy, (size_t)n, closed != 0); }
float orc_poly_perimeter(const int32_t* xy, int64_t n) { return poly_perimeter_f32(xy, (size_t)n); }
void orc_mec(const float* xy, int64_t n, float* out3) { min_enclosing_circle_f32(xy, (size_t)n, out3[0], out3[1], out3[2]); }
void orc_scale(void* in, float sx, float sy, float dx, float dy, void* out) { scale_polys(*(PolyList*)in, sx, sy, dx, dy, *(PolyList*)out); }
void orc_sort07(void* in, void* out) { sort_contours07(*(PolyList*)in, *(PolyList*)out); }
void orc_reorder(void* in, void* out, int kind) { reorder_only(*(PolyList*)in, *(PolyList*)out, kind); }
// returns number of points written (<= cap); *pass = 1 when the float32 input is passed through
int64_t orc_resample(const float* xy, int64_t n, int closed, double step, double* out, int64_t cap, int* pass) {
    std::vector<double> o; bool p;
    resample_arclen(xy, (size_t)n, closed != 0, step, o, p);
    *pass = p;
    int64_t m = (int64_t)(o.size() / 2);
    if (m <= cap && m) memcpy(out, o.data(), o.size() * sizeof(double));
    return m;
}
void orc_split_jumps(const int32_t* xy, int64_t n, double max_jump, int variant, void* out) {
    PolyList* o = (PolyList*)out; o->clear();
    if (variant == 8) split_on_long_jumps08(xy, (size_t)n, max_jump, *o); else split_on_long_jumps10(xy, (size_t)n, max_jump, *o);
}
void orc_split_small_taps08(void* in, const double* prm, void* kept, void* taps) {
    ((PolyList*)kept)->clear(); ((Taps*)taps)->xy.clear();
    split_small_and_taps08(*(PolyList*)in, p08(prm), *(PolyList*)kept, *(Taps*)taps);
}
void orc_virtual_draw08(const int32_t* xy, int64_t n, const double* prm, u8* mask, void* segs) {
    ((PolyList*)segs)->clear();
    virtual_draw08(xy, (size_t)n, p08(prm), mask, *(PolyList*)segs);
}
// groups returned as: out_group_of[i] = rank of the group of bbox i (groups ranked by smallest member)
void orc_cluster_by_overlap(const int32_t* bboxes, int n, int32_t* out_group_of) {
    std::vector<int32_t> b(bboxes, bboxes + 4 * n); std::vector<std::vector<int>> g;
    cluster_by_overlap(b, g);
    for (size_t gi = 0; gi < g.size(); gi++) for (int i : g[gi]) out_group_of[i] = (int32_t)gi;
}
int64_t orc_bfs_path(const u8* img, int h, int w, int sy, int sx, int gy, int gx, int32_t* out_yx, int64_t cap) {
    std::vector<int32_t> p; bfs_path(img, h, w, sy, sx, gy, gx, p);
    int64_t m = (int64_t)(p.size() / 2);
    if (m <= cap && m) memcpy(out_yx, p.data(), p.size() * 4);
    return m;
}
int64_t orc_component_best_path(const u8* comp, int h, int w, int has_a, int ay, int ax, int has_b, int by, int bx,
                                int min_len, int32_t* out_yx, int64_t cap) {
    std::vector<int32_t> p; component_best_path(comp, h, w, has_a, ay, ax, has_b, by, bx, min_len, p);
    int64_t m = (int64_t)(p.size() / 2);
    if (m <= cap && m) memcpy(out_yx, p.data(), p.size() * 4);
    return m;
}
void orc_post_skeleton_merge(void* lines, const double* prm, void* out) { post_skeleton_merge(*(PolyList*)lines, p08(prm), *(PolyList*)out); }
void orc_stage08(void* sorted, const double* prm, void* lines, void* taps) { stage08_layer(*(PolyList*)sorted, p08(prm), *(PolyList*)lines, *(Taps*)taps); }
void orc_cut_poly(const int32_t* xy, int64_t n, const u8* forb, int H, int W, double step, void* out) {
    ((PolyList*)out)->clear();
    cut_poly_against_mask(xy, (size_t)n, forb, H, W, step, *(PolyList*)out);
}
void orc_tiny_and_taps10(void* in, const double* prm, void* kept, void* taps) {
    ((PolyList*)kept)->clear(); ((Taps*)taps)->xy.clear();
    tiny_and_taps10(*(PolyList*)in, p10(prm), *(PolyList*)kept, *(Taps*)taps);
}
void orc_stage10_layer(void* lines_in, void* taps_in, const double* prm, u8* forbidden, void* lines_out, void* taps_out) {
    stage10_layer(*(PolyList*)lines_in, *(Taps*)taps_in, p10(prm), forbidden, *(PolyList*)lines_out, *(Taps*)taps_out);
}
// ops: 5 int32 per op (type, line_idx, flip, x, y); returns count
int64_t orc_build_ops12(void* lines, void* taps, double R, int32_t* out, int64_t cap) {
    std::vector<Op> ops; build_ops12(*(PolyList*)lines, *(Taps*)taps, R, ops);
    if ((int64_t)ops.size() <= cap)
        for (size_t i = 0; i < ops.size(); i++) { out[5 * i] = ops[i].type; out[5 * i + 1] = ops[i].line_idx; out[5 * i + 2] = ops[i].flip; out[5 * i + 3] = ops[i].x; out[5 * i + 4] = ops[i].y; }
    return (int64_t)ops.size();
}

}  // extern "C"
