// csrc/vector.hip -- stages 05 / 07 / 08 / 10 / 12 (placeholder entry points; filled in stage by stage)
#include "orip_ctx.h"
extern "C" int orip_scale_vectors(orip_ctx* c, int, float, float, float, float) { ORIP_FAIL(c, "not implemented yet"); }
extern "C" int orip_sort_contours(orip_ctx* c, int) { ORIP_FAIL(c, "not implemented yet"); }
extern "C" int orip_dedup_layer(orip_ctx* c, int, const orip_params08*) { ORIP_FAIL(c, "not implemented yet"); }
extern "C" int orip_dedup_cross(orip_ctx* c, const int32_t*, int, const orip_params10*) { ORIP_FAIL(c, "not implemented yet"); }
extern "C" int orip_plot_order(orip_ctx* c, int, double, int64_t*) { ORIP_FAIL(c, "not implemented yet"); }
extern "C" int orip_get_ops(orip_ctx* c, int, int32_t*) { ORIP_FAIL(c, "not implemented yet"); }
