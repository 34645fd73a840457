// point_tail.h — the tail of the stereo association of the key-points (src/stereoFrame.cpp:155-172): epipolar / disparity filters on
// FLOAT differences, back-projection, PointFeature sigma2, the kept left descriptor rows — one matched left key-point at a time.
// Shared by point_tail_kernel (seq_pipeline.hip) and the one-workgroup point matcher (grid_kernels.hip), which runs it as the
// last phase of a frame; the ordered compaction (ascending left index, :161-172) is the caller's.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/stvo_hip.h"

namespace stvo {

struct PointTail {
    const float* kp_l;       // [B][K][2]
    const float* kp_r;       // [B][K][2]
    const int32_t* oct_l;    // [B][K]
    const uint8_t* desc_l;   // [B][K][32]
    const stvo_cam* cams;    // [B]
    double max_dist_epip, min_disp, orb_scale_factor;
    double* pl;              // [B][K][2] out: stereo_pt pixel
    double* P;               // [B][K][3] out: back-projected point
    double* s2;              // [B][K]    out: sigma2
    uint8_t* desc;           // [B][K][32] out: pdesc_l rows
    int32_t* n;              // [B] out: stereo points of the frame
    int32_t* host_n;         // [B] or nullptr: the same count in host-visible memory
    int32_t* nl;             // [B]: zeroed when zero_nl (a frame without key-lines)
    int zero_nl;
};

// :157-160 — does left key-point i with stereo match i2 pass?  (off = b * K)
__device__ __forceinline__ bool point_tail_filter(const PointTail& t, size_t off, int i, int i2, double& disp) {
    const float yl = t.kp_l[(off + i) * 2 + 1], yr = t.kp_r[(off + i2) * 2 + 1];
    if (!((double)fabsf(yl - yr) <= t.max_dist_epip)) return false;  // float difference (:157)
    disp = (double)(t.kp_l[(off + i) * 2 + 0] - t.kp_r[(off + i2) * 2 + 0]);  // float difference (:159)
    return disp >= t.min_disp;
}

// :161-172 — left key-point i becomes stereo point k of its frame (k already includes b * K)
__device__ __forceinline__ void point_tail_write(const PointTail& t, const stvo_cam& cam, size_t off, int i, size_t k, double disp) {
    const double u = (double)t.kp_l[(off + i) * 2 + 0], v = (double)t.kp_l[(off + i) * 2 + 1];
    const double bd = cam.b / disp;  // backProjection (src/pinholeStereoCamera.cpp:221-229)
    t.pl[k * 2 + 0] = u;
    t.pl[k * 2 + 1] = v;
    t.P[k * 3 + 0] = bd * (u - cam.cx);
    t.P[k * 3 + 1] = bd * (v - cam.cy);
    t.P[k * 3 + 2] = bd * cam.fx;
    double sg = 1.0;  // PointFeature ctor: sigma2 = 1 / scale^(2 level) (src/stereoFeatures.cpp:41-47)
    const int level = t.oct_l[off + i];
    for (int q = 0; q < level; ++q) sg *= t.orb_scale_factor;
    t.s2[k] = 1.0 / (sg * sg);
    const uint4* src = reinterpret_cast<const uint4*>(t.desc_l + (off + i) * STVO_DESC_BYTES);
    uint4* dst = reinterpret_cast<uint4*>(t.desc + k * STVO_DESC_BYTES);
    dst[0] = src[0];
    dst[1] = src[1];
}

}  // namespace stvo
