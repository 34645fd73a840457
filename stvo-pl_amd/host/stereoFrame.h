// stereoFrame.h — StVO::StereoFrame with the reference's public fields (include/stereoFrame.h:92-114).
// Feature detection (detectPointFeatures / detectLineFeatures) is out of scope: the frame is built
// from FrameFeatures; the stereo association (matchStereoPoints / matchStereoLines) runs its
// descriptor matching on the GPU through the C-ABI (stvo_match_grid_points / _lines).
#pragma once
#include <vector>

#include "../../include/stvo_hip.h"
#include "pinholeStereoCamera.h"
#include "stereoFeatures.h"
#include "stvo_compat.h"

#define GRID_ROWS 48
#define GRID_COLS 64

namespace StVO {

class StereoFrame {
public:
    // ctx_lines_: optional second context (own stream / staging arena) so that the line association can run on its own
    // thread like the reference's plInParallel branch (src/stereoFrame.cpp:64-72); nullptr = everything on ctx_
    StereoFrame(const FrameFeatures& feat_, const int idx_, PinholeStereoCamera* cam_, stvo_ctx* ctx_, stvo_ctx* ctx_lines_ = nullptr);
    ~StereoFrame();

    void extractStereoFeatures(double llength_th, int fast_th = 20);
    void matchStereoPoints(std::vector<KeyPoint> points_l, std::vector<KeyPoint> points_r, DescMat& pdesc_l_,
                           DescMat pdesc_r, bool initial = false);
    void matchStereoLines(std::vector<KeyLine> lines_l, std::vector<KeyLine> lines_r, DescMat& ldesc_l_,
                          DescMat ldesc_r, bool initial = false);
    // the host halves of the two functions above (everything after matchGrid), and their use with matches that were
    // computed by the device-resident pipeline
    void buildStereoPoints(const std::vector<KeyPoint>& points_l, const std::vector<KeyPoint>& points_r, DescMat& pdesc_l_,
                           const int32_t* matches_12, bool initial);
    void buildStereoLines(const std::vector<KeyLine>& lines_l, const std::vector<KeyLine>& lines_r, DescMat& ldesc_l_,
                          const int32_t* matches_12, bool initial);
    void adoptStereoMatches(const int32_t* m12_points, const int32_t* m12_lines);
    void filterLineSegmentDisparity(Vector2d spl, Vector2d epl, Vector2d spr, Vector2d epr, double& disp_s,
                                    double& disp_e);
    double lineSegmentOverlapStereo(double spl_obs, double epl_obs, double spl_proj, double epl_proj);
    double lineSegmentOverlap(Vector2d spl_obs, Vector2d epl_obs, Vector2d spl_proj, Vector2d epl_proj);

    int frame_idx;
    Matrix4d Tfw;
    Matrix4d DT;
    Matrix6d Tfw_cov;
    Matrix6d DT_cov;
    Vector6d DT_cov_eig;
    double err_norm;

    std::vector<PointFeature*> stereo_pt;
    std::vector<LineFeature*> stereo_ls;

    std::vector<KeyPoint> points_l, points_r;
    std::vector<KeyLine> lines_l, lines_r;
    DescMat pdesc_l, pdesc_r, ldesc_l, ldesc_r;

    PinholeStereoCamera* cam;
    double inv_width, inv_height;  // grid cell

private:
    stvo_ctx* ctx;
    stvo_ctx* ctx_l;  // context used by matchStereoLines (== ctx when no second context was given)
};

}  // namespace StVO
