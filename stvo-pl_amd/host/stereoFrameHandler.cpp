// stereoFrameHandler.cpp — host mirror of /root/reference/src/stereoFrameHandler.cpp:35-180,292-392.
#include "stereoFrameHandler.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <future>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../csrc/pose_math.h"

namespace StVO {

namespace {
void check(int rc, const char* what, stvo_ctx* ctx) {
    if (rc != STVO_OK)
        throw std::runtime_error(std::string("[StVO-HIP] ") + what + ": " + stvo_error_string(rc) + " " +
                                 (ctx ? stvo_ctx_last_error(ctx) : ""));
}
}  // namespace

StereoFrameHandler::StereoFrameHandler(PinholeStereoCamera* cam_, int device_id)
    : orb_fast_th(20), llength_th(0.0), prev_frame(nullptr), curr_frame(nullptr), cam(cam_), n_inliers(0),
      n_inliers_pt(0), n_inliers_ls(0), ctx(nullptr), ctx_lines(nullptr) {
    last_result = stvo_pose_result{};
    // no CPU path exists: without a gfx950 device this throws
    check(stvo_ctx_create(device_id, 8192, 1, &ctx), "stvo_ctx_create", nullptr);
    if (!std::getenv("STVO_NO_LINE_CTX")) check(stvo_ctx_create(device_id, 2048, 1, &ctx_lines), "stvo_ctx_create(lines)", nullptr);
    const char* e = std::getenv("STVO_HANDLER_PIPELINE");
    if (e && e[0] == '0') use_pipeline = false;
}

StereoFrameHandler::~StereoFrameHandler() {
    for (auto pt : matched_pt) delete pt;
    for (auto ls : matched_ls) delete ls;
    if (curr_frame && curr_frame != prev_frame) delete curr_frame;
    delete prev_frame;
    if (seq) stvo_seq_destroy(seq);
    if (orb) stvo_orb_destroy(orb);
    if (lsd) stvo_lsd_destroy(lsd);
    if (lbd) stvo_lbd_destroy(lbd);
    if (ctx_lines) stvo_ctx_destroy(ctx_lines);
    stvo_ctx_destroy(ctx);
}

// :35-52
void StereoFrameHandler::initialize(const FrameFeatures& feat, const int idx_) {
    orb_fast_th = Config::orbFastTh();
    llength_th = Config::minLineLength() * std::min(cam->getWidth(), cam->getHeight());
    prev_frame = new StereoFrame(feat, idx_, cam, ctx, ctx_lines);
    pose_pending = false;
    if (seq) {  // a new sequence starts
        stvo_seq_destroy(seq);
        seq = nullptr;
    }
    if (use_pipeline && pipelineEnqueue(feat)) {
        pipelineCollect(prev_frame);
    } else {
        use_pipeline = false;
        prev_frame->extractStereoFeatures(llength_th, orb_fast_th);
    }
    prev_frame->Tfw = Matrix4d::Identity();
    prev_frame->Tfw_cov = Matrix6d::Identity();
    prev_frame->DT = Matrix4d::Identity();
    curr_frame = prev_frame;
    kf = KeyFrameState{};  // :48-51
}

// StereoFrame::detectStereoPoints (src/stereoFrame.cpp:88-118) for the pair: both images in one batched launch chain of the ORB
// front-end (the reference runs the two detectPointFeatures calls on two threads when lrInParallel), key-points with their
// octaves and descriptors back on the host where the StereoFrame keeps them (points_l / points_r / pdesc_l / pdesc_r).
FrameFeatures StereoFrameHandler::detectStereoFeatures(const GrayImage& img_l, const GrayImage& img_r) {
    if (img_l.empty() || img_r.empty() || img_l.rows != img_r.rows || img_l.cols != img_r.cols)
        throw std::runtime_error("[StVO-HIP] detectStereoFeatures: two non-empty 8-bit images of one size expected");
    FrameFeatures feat;
    feat.img_cols = img_l.cols;
    feat.img_rows = img_l.rows;
    const size_t px = (size_t)img_l.rows * img_l.cols;
    std::vector<uint8_t> pair(2 * px);
    const GrayImage* im[2] = {&img_l, &img_r};
    for (int s = 0; s < 2; ++s) {
        const size_t step = im[s]->step ? im[s]->step : (size_t)im[s]->cols;
        for (int r = 0; r < im[s]->rows; ++r) std::memcpy(pair.data() + s * px + (size_t)r * im[s]->cols, im[s]->data + r * step, (size_t)im[s]->cols);
    }
    if (Config::hasLines()) detectStereoLines(pair.data(), img_l.cols, img_l.rows, feat);  // src/stereoFrame.cpp:191-203
    if (!Config::hasPoints()) return feat;  // src/stereoFrame.cpp:106
    const int K = 4096;  // capacity per image: orb_nfeatures plus the ties at the cut
    // fast_th == 0 means "the configured threshold" (src/stereoFrame.cpp:109-112), otherwise the adaptive one; cv::FAST's range
    const int fast_th = std::min(254, std::max(1, orb_fast_th == 0 ? Config::orbFastTh() : orb_fast_th));
    if (orb && (orb_cols != img_l.cols || orb_rows != img_l.rows)) {
        stvo_orb_destroy(orb);
        orb = nullptr;
    }
    if (!orb) {
        stvo_orb_params prm{};
        prm.nfeatures = Config::orbNFeatures();
        prm.fast_threshold = fast_th;
        prm.edge_threshold = Config::orbEdgeTh();
        prm.nlevels = Config::orbNLevels();
        prm.scale_factor = Config::orbScaleFactor();
        check(stvo_orb_create(ctx, 2, img_l.cols, img_l.rows, K, &prm, &orb), "stvo_orb_create", ctx);
        orb_cols = img_l.cols;
        orb_rows = img_l.rows;
    }
    check(stvo_orb_set_fast_threshold(orb, fast_th), "stvo_orb_set_fast_threshold", ctx);
    std::vector<float> kp((size_t)2 * K * 2), resp((size_t)2 * K), ang((size_t)2 * K);
    std::vector<int32_t> oct((size_t)2 * K);
    std::vector<uint8_t> desc((size_t)2 * K * 32);
    int32_t n[2] = {0, 0}, nt[2] = {0, 0};
    check(stvo_orb_detect_levels(orb, pair.data(), kp.data(), resp.data(), ang.data(), oct.data(), desc.data(), n, nt), "stvo_orb_detect_levels", ctx);
    for (int s = 0; s < 2; ++s) {
        if (nt[s] > n[s])  // more key-points qualified than the front-end's per-image capacity: not silently
            std::fprintf(stderr, "[StVO-HIP] detectStereoFeatures: %d of %d key-points of the %s image dropped at the capacity of %d\n",
                         nt[s] - n[s], nt[s], s ? "right" : "left", K);
        std::vector<KeyPoint>& pts = s ? feat.points_r : feat.points_l;
        DescMat& dm = s ? feat.pdesc_r : feat.pdesc_l;
        pts.reserve(n[s]);
        for (int i = 0; i < n[s]; ++i) {
            const size_t k = (size_t)s * K + i;
            pts.push_back(KeyPoint{kp[2 * k], kp[2 * k + 1], oct[k]});
            dm.push_back_row(desc.data() + k * 32);
        }
    }
    return feat;
}

// StereoFrame::detectStereoLineSegments (src/stereoFrame.cpp:191-203) -> detectLineFeatures (:207-243) for both images of the
// pair in one batch: the LSD detector with Config's lsd_* options and min_line_length x min(cols, rows) (stvo_lsd_*, oracle/
// stvo_lsd_oracle.c), the top-N cut by response, then the LBD descriptors (stvo_lbd_*).  The FLD branch (:245-303,
// cv::ximgproc::FastLineDetector) is not built: use_fld_lines = true is refused.
void StereoFrameHandler::detectStereoLines(const uint8_t* pair, int cols, int rows, FrameFeatures& feat) {
    if (Config::useFLDLines()) throw std::runtime_error("[StVO-HIP] use_fld_lines: the FLD detector is not built (LSD only)");
    const int M = STVO_POSE_MAX_LINES;  // the pipeline's capacity per image; lsd_nfeatures (300 / 100) lies below it
    if (lsd && (line_cols != cols || line_rows != rows)) {
        stvo_lsd_destroy(lsd); lsd = nullptr;
        stvo_lbd_destroy(lbd); lbd = nullptr;
    }
    if (!lsd) {
        stvo_lsd_params prm{};
        prm.refine = Config::lsdRefine(); prm.n_bins = Config::lsdNBins(); prm.scale = Config::lsdScale();
        prm.sigma_scale = Config::lsdSigmaScale(); prm.quant = Config::lsdQuant(); prm.ang_th = Config::lsdAngTh();
        prm.log_eps = Config::lsdLogEps(); prm.density_th = Config::lsdDensityTh();
        prm.min_length = Config::minLineLength() * std::min(cols, rows);  // llength_th (:52) for this image size
        prm.nfeatures = Config::lsdNFeatures();
        check(stvo_lsd_create(ctx, 2, cols, rows, M, &prm, &lsd), "stvo_lsd_create", ctx);
        check(stvo_lbd_create(ctx, 2, cols, rows, M, &lbd), "stvo_lbd_create", ctx);
        line_cols = cols;
        line_rows = rows;
    }
    std::vector<stvo_keyline> kl((size_t)2 * M);
    std::vector<uint8_t> desc((size_t)2 * M * 32);
    int32_t n[2] = {0, 0};
    check(stvo_lsd_detect(lsd, pair, kl.data(), nullptr, n), "stvo_lsd_detect", ctx);
    {   // "keep all" (lsd_nfeatures = 0) or a budget above the pipeline's capacity: say so when the capacity cut the lines
        int32_t n_seg[2] = {0, 0}, n_pass[2] = {0, 0};
        check(stvo_lsd_counts(lsd, n_seg, n_pass), "stvo_lsd_counts", ctx);
        const int want = Config::lsdNFeatures();
        for (int s = 0; s < 2; ++s)
            if (n_seg[s] > 8192 || (n_pass[s] > M && (want == 0 || want > M)))
                std::cout << "[StVO-HIP] " << (s ? "right" : "left") << " image: " << n_seg[s] << " segments, " << n_pass[s]
                          << " longer than min_line_length; the strongest " << n[s] << " are kept (capacity " << M << " key-lines, 8192 ranked segments)"
                          << std::endl;
    }
    check(stvo_lbd_compute(lbd, pair, kl.data(), n, desc.data(), nullptr), "stvo_lbd_compute", ctx);
    for (int s = 0; s < 2; ++s) {
        std::vector<KeyLine>& ls = s ? feat.lines_r : feat.lines_l;
        DescMat& dm = s ? feat.ldesc_r : feat.ldesc_l;
        ls.reserve(n[s]);
        for (int i = 0; i < n[s]; ++i) {
            const stvo_keyline& q = kl[(size_t)s * M + i];
            ls.push_back(KeyLine{q.sx, q.sy, q.ex, q.ey, q.angle, 0});  // one octave: octaveScale = 1 (LSDDetector_custom.cpp:257)
            dm.push_back_row(desc.data() + ((size_t)s * M + i) * 32);
        }
    }
}

void StereoFrameHandler::initialize(const GrayImage& img_l, const GrayImage& img_r, const int idx_) {
    orb_fast_th = Config::orbFastTh();  // :37 (before the detection, which takes it)
    initialize(detectStereoFeatures(img_l, img_r), idx_);
}

void StereoFrameHandler::insertStereoPair(const GrayImage& img_l, const GrayImage& img_r, const int idx_) {
    insertStereoPair(detectStereoFeatures(img_l, img_r), idx_);
}

// :54-60
void StereoFrameHandler::insertStereoPair(const FrameFeatures& feat, const int idx_) {
    using clk = std::chrono::high_resolution_clock;
    const auto t0 = clk::now();
    // device pipeline: the kernel chain of the frame is enqueued BEFORE the host-side frame object (a copy of ~180 KB of features)
    // is built — the copy then overlaps the GPU's stereo stage instead of delaying its start
    const bool enqueued = use_pipeline && pipelineEnqueue(feat);
    curr_frame = new StereoFrame(feat, idx_, cam, ctx, ctx_lines);
    if (enqueued) {
        pipelineCollect(curr_frame);
        t_stereo_ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        t_f2f_ms = 0.0;
        return;
    }
    // the frame exceeds the pipeline's capacity: the host-side lists are complete, so the per-call path can take over
    // from here (and keeps the sequence: the device-side state would be stale after this frame)
    use_pipeline = false;
    curr_frame->extractStereoFeatures(llength_th, orb_fast_th);
    const auto t1 = clk::now();
    f2fTracking();
    t_stereo_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    t_f2f_ms = std::chrono::duration<double, std::milli>(clk::now() - t1).count();
}

// The adaptive FAST threshold of updateFrame (:66-86) as a decision table: the first rule that applies moves the threshold
// by `steps` increments, clipped at the side it moves towards.  A lost frame (DT == I or err above fast_err_th) counts as
// "fewer than feat_th inliers"; the last row can never fire (the row above shadows it), upstream as here.
static int adaptFastThreshold(int th, bool lost, int n_inliers_pt) {
    struct Rule {
        int mult;   // compared with mult * fast_feat_th
        bool below; // applies when n_inliers_pt is below (true) / above (false) that figure
        int steps;  // increments of fast_inc_th
    };
    static const Rule rules[] = {{1, true, -2}, {2, true, -1}, {3, false, +1}, {4, false, +2}};
    int steps = lost ? -2 : 0;
    if (!lost)
        for (const Rule& r : rules) {
            const int ref = r.mult * Config::fastFeatTh();
            if (r.below ? n_inliers_pt < ref : n_inliers_pt > ref) {
                steps = r.steps;
                break;
            }
        }
    const int moved = th + steps * Config::fastIncTh();
    if (steps < 0) return std::max(Config::fastMinTh(), moved);
    if (steps > 0) return std::min(Config::fastMaxTh(), moved);
    return th;
}

// :62-102
void StereoFrameHandler::updateFrame() {
    if (Config::adaptativeFAST()) {
        const float err_th = Config::fastErrTh();
        const bool lost = curr_frame->DT == Matrix4d::Identity() || curr_frame->err_norm > err_th;
        orb_fast_th = adaptFastThreshold(orb_fast_th, lost, n_inliers_pt);
    }
    for (auto pt : matched_pt) delete pt;
    for (auto ls : matched_ls) delete ls;
    matched_pt.clear();
    matched_ls.clear();
    if (prev_frame != curr_frame) delete prev_frame;
    prev_frame = curr_frame;
    curr_frame = nullptr;
}

// :106-129 — like the reference, points || lines on two threads when plInParallel (each with its own GPU stream)
void StereoFrameHandler::f2fTracking() {
    for (auto pt : matched_pt) delete pt;
    for (auto ls : matched_ls) delete ls;
    matched_pt.clear();
    matched_ls.clear();
    if (ctx_lines && Config::plInParallel() && Config::hasPoints() && Config::hasLines() && !curr_frame->stereo_ls.empty() &&
        !prev_frame->stereo_ls.empty()) {  // :115-118 — two tasks, each on its own context
        auto lines = std::async(std::launch::async, [&] { matchF2FLines(); });
        matchF2FPoints();
        lines.get();
    } else {
        if (Config::hasPoints()) matchF2FPoints();
        if (Config::hasLines()) matchF2FLines();
    }
    n_inliers_pt = (int)matched_pt.size();
    n_inliers_ls = (int)matched_ls.size();
    n_inliers = n_inliers_pt + n_inliers_ls;
}

// :131-153
void StereoFrameHandler::matchF2FPoints() {
    matched_pt.clear();
    if (!Config::hasPoints() || curr_frame->stereo_pt.empty() || prev_frame->stereo_pt.empty()) return;
    std::vector<int32_t> matches_12(prev_frame->pdesc_l.rows);
    check(stvo_match_nnr_mutual(ctx, prev_frame->pdesc_l.ptr(), prev_frame->pdesc_l.rows, curr_frame->pdesc_l.ptr(),
                                curr_frame->pdesc_l.rows, (float)Config::minRatio12P(), Config::bestLRMatches() ? 1 : 0,
                                matches_12.data(), nullptr),
          "stvo_match_nnr_mutual(points)", ctx);
    buildMatchedPoints(matches_12.data(), matches_12.size());
}

// :144-152
void StereoFrameHandler::buildMatchedPoints(const int32_t* matches_12, size_t n) {
    for (size_t i1 = 0; i1 < n; ++i1) {
        const int i2 = matches_12[i1];
        if (i2 < 0) continue;
        prev_frame->stereo_pt[i1]->pl_obs = curr_frame->stereo_pt[i2]->pl;
        prev_frame->stereo_pt[i1]->inlier = true;
        matched_pt.push_back(prev_frame->stereo_pt[i1]->safeCopy());
        curr_frame->stereo_pt[i2]->idx = prev_frame->stereo_pt[i1]->idx;  // prev idx
    }
}

// :155-180
void StereoFrameHandler::matchF2FLines() {
    matched_ls.clear();
    if (!Config::hasLines() || curr_frame->stereo_ls.empty() || prev_frame->stereo_ls.empty()) return;
    std::vector<int32_t> matches_12(prev_frame->ldesc_l.rows);
    stvo_ctx* cl = ctx_lines ? ctx_lines : ctx;
    check(stvo_match_nnr_mutual(cl, prev_frame->ldesc_l.ptr(), prev_frame->ldesc_l.rows, curr_frame->ldesc_l.ptr(),
                                curr_frame->ldesc_l.rows, (float)Config::minRatio12L(), Config::bestLRMatches() ? 1 : 0,
                                matches_12.data(), nullptr),
          "stvo_match_nnr_mutual(lines)", cl);
    buildMatchedLines(matches_12.data(), matches_12.size());
}

// :167-179
void StereoFrameHandler::buildMatchedLines(const int32_t* matches_12, size_t n) {
    for (size_t i1 = 0; i1 < n; ++i1) {
        const int i2 = matches_12[i1];
        if (i2 < 0) continue;
        LineFeature* p = prev_frame->stereo_ls[i1];
        const LineFeature* c = curr_frame->stereo_ls[i2];
        p->sdisp_obs = c->sdisp;
        p->edisp_obs = c->edisp;
        p->spl_obs = c->spl;
        p->epl_obs = c->epl;
        p->le_obs = c->le;
        p->inlier = true;
        matched_ls.push_back(p->safeCopy());
        curr_frame->stereo_ls[i2]->idx = p->idx;
    }
}

// :292-305
bool StereoFrameHandler::isGoodSolution(Matrix4d DT, Matrix6d DTcov, double err) {
    double w[6];
    pm::eig6(DTcov.m, w);
    if (!pm::is_good_solution(DT.m, w, err)) {
        std::cout << std::endl << w[0] << "\t" << w[5] << "\t" << err << std::endl;
        return false;
    }
    return true;
}

// :307-392 — the optimisation itself (:332-370 and the goodness test of :372) runs on the GPU
void StereoFrameHandler::optimizePose() {
    const auto t_begin = std::chrono::high_resolution_clock::now();
    if (use_pipeline) {
        // the optimisation was enqueued by insertStereoPair right behind the f2f matching; collect it
        if (!pose_pending) throw std::runtime_error("[StVO-HIP] optimizePose() without a preceding insertStereoPair()");
        pose_pending = false;
        // :317-324 — the device took this decision when it committed the previous pair; the host repeats the test for the
        // reference's console line only (isGoodSolution prints the eigenvalues of a rejected prior), while the GPU still works
        if (Config::useMotionModel()) (void)isGoodSolution(prev_frame->DT, prev_frame->DT_cov, prev_frame->err_norm);
        int32_t counts[4];
        check(stvo_seq_read(seq, &last_result, counts), "stvo_seq_read", ctx);
        const int32_t *ip = nullptr, *il = nullptr;
        check(stvo_seq_fetch_inliers(seq, &ip, &il), "stvo_seq_fetch_inliers", ctx);
        // inlier flags of the MATCHED prev features, in list order (= ascending prev stereo index)
        size_t k = 0;
        for (auto pt : matched_pt) {
            while (k < prev_frame->stereo_pt.size() && ip[k] < 0) ++k;
            pt->inlier = (k < prev_frame->stereo_pt.size()) ? ip[k] != 0 : false;
            ++k;
        }
        k = 0;
        for (auto ls : matched_ls) {
            while (k < prev_frame->stereo_ls.size() && il[k] < 0) ++k;
            ls->inlier = (k < prev_frame->stereo_ls.size()) ? il[k] != 0 : false;
            ++k;
        }
        publishPose();
        t_pose_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_begin).count();
        return;
    }
    Matrix4d DT;
    if (Config::useMotionModel()) {  // :317-324
        DT = prev_frame->DT;
        if (!isGoodSolution(DT, prev_frame->DT_cov, prev_frame->err_norm)) DT = Matrix4d::Identity();
    } else
        DT = Matrix4d::Identity();

    // matched_pt / matched_ls (std::list of records) -> structure-of-arrays for the C-ABI
    const int np = (int)matched_pt.size(), nl = (int)matched_ls.size();
    std::vector<double> P(3 * np + 1), obs(2 * np + 1), s2p(np + 1), sP(3 * nl + 1), eP(3 * nl + 1), le(3 * nl + 1),
        spl(2 * nl + 1), epl(2 * nl + 1), s2l(nl + 1);
    std::vector<int32_t> ip(np + 1), il(nl + 1);
    int k = 0;
    for (auto pt : matched_pt) {
        for (int c = 0; c < 3; ++c) P[3 * k + c] = pt->P(c);
        obs[2 * k] = pt->pl_obs(0);
        obs[2 * k + 1] = pt->pl_obs(1);
        s2p[k] = pt->sigma2;
        ip[k] = pt->inlier ? 1 : 0;
        ++k;
    }
    k = 0;
    for (auto ls : matched_ls) {
        for (int c = 0; c < 3; ++c) {
            sP[3 * k + c] = ls->sP(c);
            eP[3 * k + c] = ls->eP(c);
            le[3 * k + c] = ls->le_obs(c);
        }
        for (int c = 0; c < 2; ++c) {
            spl[2 * k + c] = ls->spl(c);
            epl[2 * k + c] = ls->epl(c);
        }
        s2l[k] = ls->sigma2;
        il[k] = ls->inlier ? 1 : 0;
        ++k;
    }
    stvo_matched m{np, P.data(), obs.data(), s2p.data(), ip.data(), nl, sP.data(), eP.data(), le.data(),
                   spl.data(), epl.data(), s2l.data(), il.data()};
    stvo_opt_params prm{};
    prm.mode = mode;
    prm.has_points = Config::hasPoints();
    prm.has_lines = Config::hasLines();
    prm.min_features = Config::minFeatures();
    prm.max_iters = Config::maxIters();
    prm.max_iters_ref = Config::maxItersRef();
    prm.homog_th = Config::homogTh();
    prm.min_error = Config::minError();
    prm.min_error_change = Config::minErrorChange();
    prm.inlier_k = Config::inlierK();
    const stvo_cam c = cam->abi();
    check(stvo_optimize_pose(ctx, DT.m, &c, &prm, &m, &last_result), "stvo_optimize_pose", ctx);

    // inlier flags and counters back into the records (removeOutliers, :988-1067)
    k = 0;
    for (auto pt : matched_pt) pt->inlier = ip[k++] != 0;
    k = 0;
    for (auto ls : matched_ls) ls->inlier = il[k++] != 0;
    publishPose();
    t_pose_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_begin).count();
}

// counters, console notes and the "set estimated pose" block of optimizePose (:372-391) from last_result
void StereoFrameHandler::publishPose() {
    n_inliers_pt = last_result.n_inliers_pt;
    n_inliers_ls = last_result.n_inliers_ls;
    n_inliers = n_inliers_pt + n_inliers_ls;
    if (last_result.status == STVO_POSE_FEW_INLIERS_AFTER)
        std::cout << "[StVO] not enough inliers (after removal)" << std::endl;
    if (last_result.status == STVO_POSE_FEW_INLIERS_BEFORE)
        std::cout << "[StVO] not enough inliers (before optimization)" << std::endl;

    // set estimated pose (:372-391)
    if (last_result.status == STVO_POSE_OK) {
        for (int i = 0; i < 16; ++i) curr_frame->DT.m[i] = last_result.T[i];
        for (int i = 0; i < 36; ++i) curr_frame->DT_cov.m[i] = last_result.cov[i];
        curr_frame->err_norm = last_result.err;
        double prod[16], x[6];
        pm::mat4_mul(prev_frame->Tfw.m, curr_frame->DT.m, prod);
        pm::logmap_se3(prod, x);
        pm::expmap_se3(x, curr_frame->Tfw.m);  // :377
        pm::unccomp_se3(prev_frame->Tfw.m, prev_frame->Tfw_cov.m, curr_frame->DT_cov.m, curr_frame->Tfw_cov.m);
        for (int i = 0; i < 6; ++i) curr_frame->DT_cov_eig(i) = last_result.cov_eig[i];
    } else {
        curr_frame->DT = Matrix4d::Identity();
        curr_frame->DT_cov = Matrix6d::Zero();
        curr_frame->err_norm = -1.0;
        curr_frame->Tfw = prev_frame->Tfw;
        curr_frame->Tfw_cov = prev_frame->Tfw_cov;
        for (int i = 0; i < 6; ++i) curr_frame->DT_cov_eig(i) = 0.0;
    }
}

// One frame through the device-resident pipeline: upload, enqueue stereo association -> f2f -> pose, then rebuild the
// host-side lists from the fetched match indices while the pose kernel runs.  false = the frame does not fit.
// first half of a frame on the device-resident pipeline: features -> pinned staging -> the whole kernel chain enqueued.  Nothing here
// needs the StereoFrame object: the caller builds it (it copies ~180 KB of features) while the GPU already works
bool StereoFrameHandler::pipelineEnqueue(const FrameFeatures& feat) {
    const int n0 = (int)feat.points_l.size(), n1 = (int)feat.points_r.size(), n2 = (int)feat.lines_l.size(),
              n3 = (int)feat.lines_r.size();
    if (n0 > STVO_POSE_MAX_POINTS || n1 > STVO_POSE_MAX_POINTS || n2 > STVO_POSE_MAX_LINES || n3 > STVO_POSE_MAX_LINES) return false;
    if (!seq) {
        stvo_match_params mp{};
        mp.best_lr_matches = Config::bestLRMatches(); mp.matching_s_ws = Config::matchingSWs();
        mp.min_ratio_12_p = (float)Config::minRatio12P(); mp.min_ratio_12_l = (float)Config::minRatio12L();
        mp.max_dist_epip = Config::maxDistEpip(); mp.min_disp = Config::minDisp(); mp.line_sim_th = Config::lineSimTh();
        mp.stereo_overlap_th = Config::stereoOverlapTh(); mp.line_horiz_th = Config::lineHorizTh();
        mp.ls_min_disp_ratio = Config::lsMinDispRatio(); mp.orb_scale_factor = Config::orbScaleFactor();
        mp.lsd_scale = Config::lsdScale();
        mp.min_ratio_12_p_d = Config::minRatio12P();  // matchGrid compares with the double
        stvo_opt_params op{};
        op.mode = mode; op.has_points = Config::hasPoints(); op.has_lines = Config::hasLines();
        op.min_features = Config::minFeatures(); op.max_iters = Config::maxIters(); op.max_iters_ref = Config::maxItersRef();
        op.homog_th = Config::homogTh(); op.min_error = Config::minError(); op.min_error_change = Config::minErrorChange();
        op.inlier_k = Config::inlierK();
        const stvo_cam c = cam->abi();
        check(stvo_seq_create(ctx, 1, STVO_POSE_MAX_POINTS, STVO_POSE_MAX_LINES, feat.img_cols, feat.img_rows, &c, &mp, &op, &seq),
              "stvo_seq_create", ctx);
        check(stvo_seq_enable_fetch(seq, 1), "stvo_seq_enable_fetch", ctx);
        // :317-324 — under the motion model the device keeps the committed increment and applies the isGoodSolution rule itself
        if (Config::useMotionModel()) check(stvo_seq_set_motion_model(seq, 1), "stvo_seq_set_motion_model", ctx);
        int32_t K = 0, M = 0;
        check(stvo_seq_strides(seq, &K, &M), "stvo_seq_strides", ctx);
        seq_K = K; seq_M = M;
    }
    // cv::KeyPoint / KeyLine arrays -> the plain arrays of stvo_frame_features
    static thread_local std::vector<float> kpl, kpr, kll, klr;
    static thread_local std::vector<int32_t> ol, oll;
    kpl.resize(2 * n0 + 2); kpr.resize(2 * n1 + 2); kll.resize(4 * n2 + 4); klr.resize(4 * n3 + 4);
    ol.resize(n0 + 1); oll.resize(n2 + 1);
    for (int i = 0; i < n0; ++i) { kpl[2 * i] = feat.points_l[i].x; kpl[2 * i + 1] = feat.points_l[i].y; ol[i] = feat.points_l[i].octave; }
    for (int i = 0; i < n1; ++i) { kpr[2 * i] = feat.points_r[i].x; kpr[2 * i + 1] = feat.points_r[i].y; }
    for (int i = 0; i < n2; ++i) {
        kll[4 * i] = feat.lines_l[i].startPointX; kll[4 * i + 1] = feat.lines_l[i].startPointY;
        kll[4 * i + 2] = feat.lines_l[i].endPointX; kll[4 * i + 3] = feat.lines_l[i].endPointY;
        oll[i] = feat.lines_l[i].octave;
    }
    for (int i = 0; i < n3; ++i) {
        klr[4 * i] = feat.lines_r[i].startPointX; klr[4 * i + 1] = feat.lines_r[i].startPointY;
        klr[4 * i + 2] = feat.lines_r[i].endPointX; klr[4 * i + 3] = feat.lines_r[i].endPointY;
    }
    const int32_t n[4] = {n0, n1, n2, n3};
    stvo_frame_features ff{};
    ff.stride_kp = n0 > n1 ? n0 : n1;
    ff.stride_kl = n2 > n3 ? n2 : n3;
    ff.n_kp_l = &n[0]; ff.n_kp_r = &n[1]; ff.n_kl_l = &n[2]; ff.n_kl_r = &n[3];
    ff.kp_l = kpl.data(); ff.oct_l = ol.data(); ff.desc_l = feat.pdesc_l.ptr(); ff.kp_r = kpr.data(); ff.desc_r = feat.pdesc_r.ptr();
    ff.kl_l = kll.data(); ff.oct_ll = oll.data(); ff.ldesc_l = feat.ldesc_l.ptr(); ff.kl_r = klr.data(); ff.ldesc_r = feat.ldesc_r.ptr();
    check(stvo_seq_upload(seq, pipe_slot, &ff), "stvo_seq_upload", ctx);
    check(stvo_seq_step_dev(seq, pipe_slot), "stvo_seq_step_dev", ctx);
    pipe_slot ^= 1;
    return true;
}

// second half: the host-side lists of the frame the GPU is working on (stereo_pt / stereo_ls, matched_pt / matched_ls) from the
// match indices, which reach pinned memory right after the f2f stage — the pose kernel is still running
void StereoFrameHandler::pipelineCollect(StereoFrame* frame) {
    const bool first = (frame == prev_frame);
    const int32_t *ms_p = nullptr, *ms_l = nullptr, *m_p = nullptr, *m_l = nullptr;
    check(stvo_seq_fetch_matches(seq, &ms_p, &ms_l, &m_p, &m_l), "stvo_seq_fetch_matches", ctx);
    frame->adoptStereoMatches(ms_p, ms_l);
    if (first) {
        int32_t counts[4];
        stvo_pose_result dummy;
        check(stvo_seq_read(seq, &dummy, counts), "stvo_seq_read", ctx);  // nothing to track on the first frame
        return;
    }
    // f2fTracking (:106-129) from the fetched f2f matches
    for (auto pt : matched_pt) delete pt;
    for (auto ls : matched_ls) delete ls;
    matched_pt.clear();
    matched_ls.clear();
    if (Config::hasPoints() && !curr_frame->stereo_pt.empty() && !prev_frame->stereo_pt.empty())
        buildMatchedPoints(m_p, prev_frame->stereo_pt.size());
    if (Config::hasLines() && !curr_frame->stereo_ls.empty() && !prev_frame->stereo_ls.empty())
        buildMatchedLines(m_l, prev_frame->stereo_ls.size());
    n_inliers_pt = (int)matched_pt.size();
    n_inliers_ls = (int)matched_ls.size();
    n_inliers = n_inliers_pt + n_inliers_ls;
    pose_pending = true;
}

void StereoFrameHandler::resetOutliers() {
    for (auto pt : matched_pt) pt->inlier = true;
    for (auto ls : matched_ls) ls->inlier = true;
    n_inliers_pt = (int)matched_pt.size();
    n_inliers_ls = (int)matched_ls.size();
    n_inliers = n_inliers_pt + n_inliers_ls;
}

// :1136-1188
bool StereoFrameHandler::needNewKF() {
    return kf_need_new(kf, curr_frame->Tfw, curr_frame->DT, curr_frame->DT_cov, Config::minEntropyRatio(), Config::maxKFTDist(),
                       Config::maxKFRDist());
}

// :1190-1218
void StereoFrameHandler::currFrameIsKF() {
    int idx_pt = 0;
    for (auto pt : curr_frame->stereo_pt) pt->idx = idx_pt++;
    int idx_ls = 0;
    for (auto ls : curr_frame->stereo_ls) ls->idx = idx_ls++;
    curr_frame->Tfw = Matrix4d::Identity();
    curr_frame->Tfw_cov = Matrix6d::Identity();
    kf_reset(kf, curr_frame->Tfw);
}

void StereoFrameHandler::setAsOutliers() {
    for (auto pt : matched_pt) pt->inlier = false;
    for (auto ls : matched_ls) ls->inlier = false;
    n_inliers_pt = n_inliers_ls = n_inliers = 0;
}

}  // namespace StVO
