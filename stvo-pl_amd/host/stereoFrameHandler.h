// stereoFrameHandler.h — StVO::StereoFrameHandler with the reference's method names and public
// fields (include/stereoFrameHandler.h:41-85).  `Mat img_l, img_r` arguments are replaced by the
// frame's extracted FrameFeatures (feature extraction is out of scope); everything downstream —
// stereo association, f2f tracking, optimizePose, updateFrame — keeps the reference's behaviour,
// with the descriptor matching and the whole pose optimisation running on the GPU via the C-ABI.
#pragma once
#include <list>

#include "../../include/stvo_hip.h"
#include "keyframe.h"
#include "stereoFrame.h"

namespace StVO {

class StereoFrameHandler {
public:
    explicit StereoFrameHandler(PinholeStereoCamera* cam_, int device_id = 0);
    ~StereoFrameHandler();

    void initialize(const FrameFeatures& feat, const int idx_);
    void insertStereoPair(const FrameFeatures& feat, const int idx_);
    // The reference's own entry points, initialize / insertStereoPair(const Mat img_l, const Mat img_r, const int idx)
    // (include/stereoFrameHandler.h:44-45): the two rectified 8-bit images go through the ORB point front-end on the GPU
    // (stvo_orb_detect_levels with Config's orb_* values and the handler's adaptive orb_fast_th — what
    // StereoFrame::detectStereoPoints does with cv::ORB, src/stereoFrame.cpp:88-118) and continue as extracted features.
    // Key-lines: with Config::hasLines() the images also go through detectStereoLines below (LSD with lsd_refine 0 or 1 + LBD on the GPU,
    // StereoFrame::detectLineFeatures, src/stereoFrame.cpp:207-243); use_fld_lines and lsd_refine 2 are refused.
    void initialize(const GrayImage& img_l, const GrayImage& img_r, const int idx_);
    void insertStereoPair(const GrayImage& img_l, const GrayImage& img_r, const int idx_);
    void detectStereoLines(const uint8_t* pair, int cols, int rows, FrameFeatures& feat);
    FrameFeatures detectStereoFeatures(const GrayImage& img_l, const GrayImage& img_r);
    void updateFrame();

    void f2fTracking();
    void matchF2FPoints();
    void matchF2FLines();

    bool isGoodSolution(Matrix4d DT, Matrix6d DTcov, double err);
    void optimizePose();
    void resetOutliers();
    void setAsOutliers();

    // slam functions (src/stereoFrameHandler.cpp:1134-1218)
    bool needNewKF();
    void currFrameIsKF();
    KeyFrameState kf;  // prev_f_iskf, entropy_first_prevKF, T_prevKF, cov_prevKF_currF, N_prevKF_currF (:81-85)

    // adaptative fast
    int orb_fast_th;
    double llength_th;

    std::list<PointFeature*> matched_pt;
    std::list<LineFeature*> matched_ls;

    StereoFrame* prev_frame;
    StereoFrame* curr_frame;
    PinholeStereoCamera* cam;

    int n_inliers, n_inliers_pt, n_inliers_ls;

    // what the GPU reported for the last optimizePose (status / path / iteration counts)
    stvo_pose_result last_result;
    // host wall-clock per stage of the last frame, ms: stereo association, f2f matching, optimizePose
    double t_stereo_ms = 0.0, t_f2f_ms = 0.0, t_pose_ms = 0.0;
    int mode = 0;  // the local constant of src/stereoFrameHandler.cpp:329 (0 GN, 1 robust GN, 2 LM)

    // Device-resident pipeline (stvo_seq_*) behind the same methods: insertStereoPair uploads the frame
    // once and enqueues stereo association -> f2f -> optimizePose on the GPU; the host-side lists (stereo_pt / stereo_ls,
    // matched_pt / matched_ls) are rebuilt from the fetched match indices WHILE the pose kernel runs, and optimizePose()
    // only collects the result.  On by default; off when Config::useMotionModel() (the pipeline starts from DT = I),
    // when a frame exceeds the pipeline's capacity (2048 key-points / 512 key-lines per image), or with
    // STVO_HANDLER_PIPELINE=0 in the environment.
    bool use_pipeline = true;

private:
    bool pipelineEnqueue(const FrameFeatures& feat);  // false: frame does not fit -> legacy path
    void pipelineCollect(StereoFrame* frame);
    void buildMatchedPoints(const int32_t* matches_12, size_t n);
    void buildMatchedLines(const int32_t* matches_12, size_t n);
    void publishPose();
    stvo_orb* orb = nullptr;  // created at the first image pair (image size, Config's orb_* values)
    int orb_cols = 0, orb_rows = 0;
    stvo_lsd* lsd = nullptr;  // the line front-end of the image entry points: LSD detector + LBD descriptor, both images per call
    stvo_lbd* lbd = nullptr;
    int line_cols = 0, line_rows = 0;
    stvo_seq* seq = nullptr;
    int seq_K = 0, seq_M = 0, pipe_slot = 0;
    bool pose_pending = false;
    stvo_ctx* ctx;
    stvo_ctx* ctx_lines;  // second context for the line tasks of the plInParallel branches (:115-118, stereoFrame.cpp:64-72)
};

}  // namespace StVO
