// grid_kernels.hip — K3 (placeholder until implemented)
#include "ctx_internal.h"
extern "C" {
int stvo_match_grid_points(stvo_ctx*, const int32_t*, const uint8_t*, int, const int32_t*, const int32_t*, const uint8_t*,
                           int, const stvo_grid_window*, double, int, int32_t*, int32_t*) { return STVO_ERR_UNSUPPORTED; }
int stvo_match_grid_lines(stvo_ctx*, const int32_t*, const uint8_t*, int, const int32_t*, const int32_t*, const uint8_t*,
                          int, const double*, const stvo_grid_window*, double, double, int, int32_t*, int32_t*) { return STVO_ERR_UNSUPPORTED; }
}
