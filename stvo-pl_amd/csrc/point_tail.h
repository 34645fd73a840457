// point_tail.h — the tail of the stereo association of the key-points (src/stereoFrame.cpp:155-172): epipolar / disparity filters on
// FLOAT differences, the stereo point's record, the kept left descriptor rows — one matched left key-point at a time.
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
    double max_dist_epip, min_disp;
    float4* rc;              // [B][K] out: the stereo point in compact form {u, v, disparity, level} — what PointFeature is built
                             // from (:152-167); P = backProjection(u, v, disparity) and sigma2 = 1 / scale^(2 level) are rebuilt
                             // where they are used (the pose kernels, kernels.h: PoseArgs::prev_rc), not stored
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

// :161-172 — left key-point i becomes stereo point k of its frame (k already includes b * K).  pl = (u, v) and the
// disparity are floats widened to double in the reference (KeyPoint::pt, a float difference); the level is the key-point's octave.
__device__ __forceinline__ void point_tail_write(const PointTail& t, size_t off, int i, size_t k, double disp) {
    t.rc[k] = make_float4(t.kp_l[(off + i) * 2 + 0], t.kp_l[(off + i) * 2 + 1], (float)disp, (float)t.oct_l[off + i]);
    const uint4* src = reinterpret_cast<const uint4*>(t.desc_l + (off + i) * STVO_DESC_BYTES);
    uint4* dst = reinterpret_cast<uint4*>(t.desc + k * STVO_DESC_BYTES);
    dst[0] = src[0];
    dst[1] = src[1];
}

}  // namespace stvo
