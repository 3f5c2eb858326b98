/* include/orip.h -- C ABI of liborip.so: the MI355X-native hot path of omnirevolve-image-processor
 * (stages 02_color_extract -> 03_edge_detect -> 04_find_contours -> 05/07 glue -> 08_dedup_layer_basic ->
 * 10_dedup_cross_basic -> 12_optimize_plot_order).
 *
 * The reference is pure Python and has NO FFI for this path: its stage API is "one script per stage,
 * artefacts on disk" (pipeline.py:66-111).  Each entry point below therefore names the reference
 * function (file:line under /root/reference/image_processor/) whose computation it replaces; the Python
 * host (omnirevolve-image-processor_amd/orip/) binds them with ctypes and re-creates the stage scripts.
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions: plain C types; every function returns 0 on success, <0 on error (text via
 * orip_last_error); one orip_ctx per device and host thread (not re-entrant); the caller owns every host
 * buffer (C-contiguous, row-major); results stay RESIDENT on the device between calls ("slots") and are
 * moved only by the explicit orip_get_ / orip_set_ calls, so the end-to-end path 02->12 touches the host
 * only for the input image and the final ops.  Variable-length results use the two-call pattern
 * (orip_polys_size then orip_get_polys).  There is no CPU fallback anywhere in the library.
 */
#ifndef ORIP_H
#define ORIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orip_ctx orip_ctx;

#define ORIP_MAX_LAYERS 16

/* polyline-list slots (per layer), named after the artefact each one mirrors */
enum {
    ORIP_SLOT_CONTOURS = 0,     /* <layer>/contours.pkl         (04:226-228) */
    ORIP_SLOT_SCALED = 1,       /* <layer>/contours_scaled.pkl  (05:124-127) */
    ORIP_SLOT_SORTED = 2,       /* <layer>/contours_sorted.pkl  (07:91-92)   */
    ORIP_SLOT_LINES_INTRA = 3,  /* <layer>/lines_intra.pkl      (08:548-549) */
    ORIP_SLOT_LINES_CROSS = 4,  /* <layer>/lines_cross.pkl      (10:200-202) */
    ORIP_SLOT_COUNT = 5
};
enum { ORIP_TAPS_INTRA = 0 /* taps_intra.pkl 08:550-551 */, ORIP_TAPS_CROSS = 1 /* taps_cross.pkl 10:203-204 */ };

/* 08:484-509 derived parameters */
typedef struct {
    double tap_diam, tap_max_dim, min_keep, tap_max_per;
    int32_t tap_max_v;
    double sample_step, tail_len_px, col_rad, grid_stride, max_jump;
    int32_t post_on, post_brush;
    double post_step, post_eps;
    int32_t post_minlen;
    int32_t W, H;          /* canvas, 08:103-113 */
    int32_t brush_forbid;
} orip_params08;

/* 10:217-229 derived parameters */
typedef struct {
    double tap_diam, min_keep, tap_max_per;
    int32_t tap_max_v;
    double max_jump, D_lines, D_taps, step_px;
    int32_t W, H;
} orip_params10;

/* ---- context ---- */
int orip_create(int device_id, orip_ctx** out);
void orip_destroy(orip_ctx* ctx);
const char* orip_last_error(orip_ctx* ctx);
int orip_sync(orip_ctx* ctx);
/* HIP-event timing of the named raster kernel since the last reset: total ms and launch count (bench.py roofline) */
int orip_prof_reset(orip_ctx* ctx);
int orip_prof_get(orip_ctx* ctx, const char* kernel, double* total_ms, int64_t* launches);
int orip_prof_enable(orip_ctx* ctx, int on);

/* ---- stage 01: 01_resize.py ---- */
/* cv2.resize(img, (newW, newH), interpolation=cv2.INTER_AREA) for shrinking (01:19; the size rule int(w * max_dimension / max(h, w)) of 01:15-18
 * stays on the host).  src u8 [H,W,cn] (host), cn 1..4; dst u8 [newH,newW,cn] (host) may be NULL.  as_image != 0 (cn == 3): the result is left as
 * the context's image as orip_set_image would leave it, so the resident chain needs no resized.png. */
int orip_resize_area(orip_ctx* ctx, const uint8_t* src, int H, int W, int cn, int newH, int newW, uint8_t* dst, int as_image);

/* ---- stage 02: 02_color_extract.py ---- */
/* upload the pixels of resized.png (BGR u8 [H,W,3]) -- replaces cv2.imread at 02:70-71 */
int orip_set_image(orip_ctx* ctx, const uint8_t* bgr, int H, int W);
/* cv2.cvtColor(BGR2LAB) (02:35) of the whole image or of the pixels idx[0..n) -> host u8 [n,3] (test hook) */
int orip_lab_of(orip_ctx* ctx, const int64_t* idx, int64_t n, uint8_t* lab_out);
/* _kmeans_lab fit part (02:39-49): Lab of the sampled pixels + cv2.kmeans(PP centres, attempts, (EPS|ITER)) */
int orip_kmeans_fit(orip_ctx* ctx, const int64_t* sample_idx, int64_t n_idx, int K, int attempts, int max_iter,
                    double eps, float* centers_out /* [K,3] in cv2.kmeans order */, double* compactness_out);
/* ---- process_colors.py (standalone label-map tool) ---- */
/* kmeans_palette (:31-46): cv2.kmeans (PP centres) over the R, G, B bytes of the sampled pixels of the image set with orip_set_image; same
 * arguments as orip_kmeans_fit, centres in R, G, B order.  The subsample (:35-39, numpy RandomState) stays on the host. */
int orip_kmeans_fit_rgb(orip_ctx* ctx, const int64_t* sample_idx, int64_t n_idx, int K, int attempts, int max_iter,
                        double eps, float* centers_out /* [K,3] */, double* compactness_out);
/* assign_labels (:69-77): index of the nearest palette colour per pixel, with the reference's int16 arithmetic (squares of differences above
 * 181 wrap) and first-minimum ties.  palette_rgb u8 [K,3].  Leaves the labels resident (orip_get_labels); labels_out (host u8 [H,W]) and
 * counts_out ([K] pixels per label) may be NULL. */
int orip_assign_palette(orip_ctx* ctx, const uint8_t* palette_rgb, int K, uint8_t* labels_out, int64_t* counts_out);

/* assignment (02:53-55) + dark->light relabel (02:120-127) + per-cluster mask + 3x3 RECT open/close (02:144-154).
 * Leaves labels u8 [H,W] (dark->light index) and K masks resident; layer l of the context = cluster l. */
int orip_extract_layers(orip_ctx* ctx, const float* centers /* [K,3] */, int K, int open_iters, int close_iters,
                        float* centers_sorted_out /* [K,3] */, int64_t* counts_out /* [K] pixels per cluster */);
int orip_get_labels(orip_ctx* ctx, uint8_t* labels_out);
int orip_get_mask(orip_ctx* ctx, int layer, uint8_t* mask_out);
/* layer sharding: keep only the listed cluster layers, compacted to local layers 0..n-1 (mask planes only) */
int orip_keep_layers(orip_ctx* ctx, const int32_t* layers, int n);
/* upload K masks [K,H,W] (stage 03 run stand-alone from mask.png files, 03:15-19) */
int orip_set_masks(orip_ctx* ctx, const uint8_t* masks, int K, int H, int W);

/* ---- stage 03: 03_edge_detect.py process_color (03:13-40), all layers in one call ---- */
int orip_detect_edges(orip_ctx* ctx, int morph_k, int open_iters, int close_iters, int gauss_k, int low, int high);
int orip_get_edges(orip_ctx* ctx, int layer, uint8_t* edges_out);
int orip_set_edges(orip_ctx* ctx, const uint8_t* edges, int K, int H, int W);

/* ---- stage 04: 04_find_contours.py vectorize_layer (04:214-230), all layers in one call ---- */
int orip_find_contours(orip_ctx* ctx);
/* Optional hint for the resident chain (no counterpart in the reference): after orip_set_image, announce that K layers will be traced so that the
 * K memo planes of stage 04 are cleared while the k-means fit runs instead of underneath stages 02 / 03.  orip_contours_prepare works without it. */
int orip_contours_reserve(orip_ctx* ctx, int K);

/* The same work split for per-layer pipelines: prepare = the part batched over the layers (thinning 04:35-99, components, walk
 * schedule); contours_layer = the walks of one layer (04:101-211), callable for different layers from different host threads.
 * prepare also enqueues every layer's walk on that layer's stream before it returns, so contours_layer normally only completes a
 * walk that is already running (a layer whose lane is held by another call at that moment is started by its contours_layer call). */
int orip_contours_prepare(orip_ctx* ctx);
int orip_contours_layer(orip_ctx* ctx, int layer);
int orip_get_skeleton(orip_ctx* ctx, int layer, uint8_t* skel_out); /* thinning_zhangsuen output (04:35-99) */

/* ---- polyline-list / tap-list slots ---- */
int orip_polys_size(orip_ctx* ctx, int slot, int layer, int64_t* n_polys, int64_t* n_points);
/* pts may be NULL: offsets only (the CONTOURS / SCALED / SORTED lists of a resident chain are held as walk records and only expanded into
 * int32 pairs when the points are asked for -- a heavy layer holds 10^8..10^9 of them) */
int orip_get_polys(orip_ctx* ctx, int slot, int layer, int64_t* off /* [n+1] */, int32_t* pts /* [n_points,2] or NULL */);
int orip_set_polys(orip_ctx* ctx, int slot, int layer, int64_t n_polys, const int64_t* off, const int32_t* pts);
int orip_taps_size(orip_ctx* ctx, int which, int layer, int64_t* n);
int orip_get_taps(orip_ctx* ctx, int which, int layer, int32_t* xy);
int orip_set_taps(orip_ctx* ctx, int which, int layer, int64_t n, const int32_t* xy);
int orip_set_layer_count(orip_ctx* ctx, int K);

/* ---- stage 05: _scale_one (05:82-96): CONTOURS -> SCALED ---- */
int orip_scale_vectors(orip_ctx* ctx, int layer, float sx, float sy, float dx, float dy);
/* ---- stage 07: reorder_one_color (07:19-95): SCALED -> SORTED ---- */
int orip_sort_contours(orip_ctx* ctx, int layer);
/* ---- stage 08: process_layer (08:484-557): SORTED -> LINES_INTRA + TAPS_INTRA ---- */
int orip_dedup_layer(orip_ctx* ctx, int layer, const orip_params08* prm);
/* orip_contours_layer + orip_scale_vectors [+ orip_sort_contours [+ orip_dedup_layer]] of one layer in one call (upto = 5, 7 or 8; prm may be NULL
 * below 8): the per-layer front of a resident chain without returning to the caller between the stages.  Same lane rules as the single calls. */
int orip_layer_front(orip_ctx* ctx, int layer, float sx, float sy, float dx, float dy, int upto, const orip_params08* prm);
/* ---- stage 10: main (10:212-278): LINES/TAPS_INTRA -> LINES/TAPS_CROSS, layers visited in `order` ---- */
int orip_dedup_cross(orip_ctx* ctx, const int32_t* order, int n_layers, const orip_params10* prm);
/* The same loop one layer at a time (10:230-262): begin clears the cumulative raster, then the layers must be passed in `order`. */
int orip_dedup_cross_begin(orip_ctx* ctx, const orip_params10* prm);
int orip_dedup_cross_layer(orip_ctx* ctx, int layer);
/* same, reading LINES/TAPS_INTRA of slot `src_layer` and writing LINES/TAPS_CROSS of `layer` (layer-sharded processes keep their
 * own layers under local indices and stage remote ones in a spare slot) */
int orip_dedup_cross_layer_from(orip_ctx* ctx, int src_layer, int layer);
/* same, but the travel reorder of the kept lines (10:253) is left to orip_plot_order(layer), which runs it first on the layer's own
 * stream: LINES_CROSS of `layer` is in cut order until then (resident pipelines only; nothing else in stage 10 depends on that order) */
int orip_dedup_cross_layer_deferred(orip_ctx* ctx, int src_layer, int layer);
/* 1 when the library was built with the replaced kernel variants (`make variants` -> liborip_variants.so, -DORIP_VARIANTS: the one-workgroup k-means
 * fit, byte-plane thinning, the first grid greedy kernel, the sequential tail simulation, the serial cumulative-length chain; selected through
 * their ORIP_* environment switches for agreement tests), 0 for the default library, where those switches read as not set. */
int orip_has_variants(void);
/* ---- previews 06 / 09 / 11 (06_preview_scaled.py:76-88 _draw_layer, 09_preview_intra.py:71-88 _draw_lines / _draw_taps, 11_preview_cross.py) ----
 * Coverage planes (0 = untouched .. 255 = fully covered, H*W bytes each, host buffers, either may be NULL) of the polylines of (slot, layer) drawn
 * `thickness` px wide and of the taps `taps_which` (ORIP_TAPS_INTRA / ORIP_TAPS_CROSS, -1: none) as filled discs of `radius` px.  antialias != 0: the
 * coverage falls off linearly over one pixel around the outline.  This stands in for cv2.LINE_AA, which the reference's tests do not pin: PARITY
 * UNPINNED, visual QA only (csrc/vector_preview.hip states what is drawn; oracle/oracle.py: preview_cover is the same, bit for bit). */
int orip_preview_cover(orip_ctx* ctx, int slot, int layer, int taps_which, int W, int H, int thickness, int radius, int antialias, uint8_t* line_cov, uint8_t* tap_cov);
/* ---- stage 12: _build_ops_for_layer (12:85-187): LINES/TAPS_CROSS -> ops ----
 * ops are returned as 5 int32 each: (type 0 line / 1 tap, line index into LINES_CROSS, flip, x, y). */
int orip_plot_order(orip_ctx* ctx, int layer, double R_insert, int64_t* n_ops);
int orip_get_ops(orip_ctx* ctx, int layer, int32_t* ops5);

/* ---- after the path: 13_build_stream.py ----
 * Direction codes of n moves (x0, y0, x1, y1 in plotter steps), the helper's bresenham_dir_codes (shared/omnirevolve_plotter_stream_creator_helper.py
 * :183-207) for all pen-up travels and polyline segments of a plot at once: codes 0 +Y, 1 NE, 2 +X, 3 SE, 4 -Y, 5 SW, 6 -X, 7 NW, concatenated in
 * move order.  Two-call pattern: orip_stream_codes leaves them resident and returns the total, the fetch copies off[n+1] and codes[total]. */
int orip_stream_codes(orip_ctx* ctx, const int32_t* moves /* [n,4] */, int64_t n, int64_t* total_steps);
int orip_stream_codes_fetch(orip_ctx* ctx, int64_t* off_out /* [n+1] */, uint8_t* codes_out /* [total] */);

/* ---- multi-GPU exchange (no counterpart in the reference, which is a single process) ----
 * One process per GPU; rank r owns the cluster layers {l : l % world == r} for stages 03-08 and 12.  Stage 10 is replicated and needs
 * every layer's stage-08 lists (10:236-267): orip_bcast_layer sends LINES_INTRA / TAPS_INTRA of one layer from its owner to all ranks
 * with RCCL broadcasts over xGMI, device to device.  The unique id is created on rank 0 and handed to the other processes by the
 * launcher (any out-of-band channel: bench.py uses its torch.distributed store). */
#define ORIP_COMM_ID_BYTES 128
int orip_comm_unique_id(uint8_t* id_out /* [ORIP_COMM_ID_BYTES] */);
int orip_comm_init(orip_ctx* ctx, const uint8_t* id, int rank, int world);
int orip_comm_destroy(orip_ctx* ctx);
/* collective: every rank passes the slot index under which IT holds / receives the layer */
int orip_bcast_layer(orip_ctx* ctx, int root, int my_slot);

#ifdef __cplusplus
}
#endif
#endif
