// fast_atan2.h — cv::fastAtan2 as the image front-ends evaluate it (orb_kernels.hip: the intensity-centroid angle; lsd_kernels.hip:
// level-line angles, region angles).  FP32, no fused multiply-adds.
#pragma once
#include <hip/hip_runtime.h>

// no fused multiply-adds: OpenCV evaluates the polynomial with separate multiplications and additions (the oracles are built with
// -ffp-contract=off).  File-scope pragma: it also covers whatever follows the #include — both including files want exactly that.
#pragma clang fp contract(off)

namespace stvo {

// OpenCV fastAtan2 (degrees): 7th-order odd polynomial on [0, 1], octant folding
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale, p5 = 0.1555786518463281f * scale,
                p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    // one division for both octant cases (lanes of a wave take both: as two branches every lane paid for two divisions)
    const bool x_major = ax >= ay;
    const float c = __fdiv_rn(x_major ? ay : ax, (x_major ? ax : ay) + (float)2.2204460492503131e-16);
    const float c2 = c * c;
    float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    if (!x_major) a = 90.f - a;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

}  // namespace stvo
