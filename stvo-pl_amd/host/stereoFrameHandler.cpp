// stereoFrameHandler.cpp — host mirror of /root/reference/src/stereoFrameHandler.cpp:35-180,292-392.
#include "stereoFrameHandler.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
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
}

StereoFrameHandler::~StereoFrameHandler() {
    for (auto pt : matched_pt) delete pt;
    for (auto ls : matched_ls) delete ls;
    if (curr_frame && curr_frame != prev_frame) delete curr_frame;
    delete prev_frame;
    if (ctx_lines) stvo_ctx_destroy(ctx_lines);
    stvo_ctx_destroy(ctx);
}

// :35-52
void StereoFrameHandler::initialize(const FrameFeatures& feat, const int idx_) {
    orb_fast_th = Config::orbFastTh();
    llength_th = Config::minLineLength() * std::min(cam->getWidth(), cam->getHeight());
    prev_frame = new StereoFrame(feat, idx_, cam, ctx, ctx_lines);
    prev_frame->extractStereoFeatures(llength_th, orb_fast_th);
    prev_frame->Tfw = Matrix4d::Identity();
    prev_frame->Tfw_cov = Matrix6d::Identity();
    prev_frame->DT = Matrix4d::Identity();
    curr_frame = prev_frame;
    kf = KeyFrameState{};  // :48-51
}

// :54-60
void StereoFrameHandler::insertStereoPair(const FrameFeatures& feat, const int idx_) {
    using clk = std::chrono::high_resolution_clock;
    const auto t0 = clk::now();
    curr_frame = new StereoFrame(feat, idx_, cam, ctx, ctx_lines);
    curr_frame->extractStereoFeatures(llength_th, orb_fast_th);
    const auto t1 = clk::now();
    f2fTracking();
    t_stereo_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    t_f2f_ms = std::chrono::duration<double, std::milli>(clk::now() - t1).count();
}

// :62-102
void StereoFrameHandler::updateFrame() {
    if (Config::adaptativeFAST()) {
        const int min_fast = Config::fastMinTh(), max_fast = Config::fastMaxTh(), fast_inc = Config::fastIncTh(),
                  feat_th = Config::fastFeatTh();
        const float err_th = Config::fastErrTh();
        if (curr_frame->DT == Matrix4d::Identity() || curr_frame->err_norm > err_th)
            orb_fast_th = std::max(min_fast, orb_fast_th - 2 * fast_inc);
        else if (n_inliers_pt < feat_th)
            orb_fast_th = std::max(min_fast, orb_fast_th - 2 * fast_inc);
        else if (n_inliers_pt < feat_th * 2)
            orb_fast_th = std::max(min_fast, orb_fast_th - fast_inc);
        else if (n_inliers_pt > feat_th * 3)
            orb_fast_th = std::min(max_fast, orb_fast_th + fast_inc);
        else if (n_inliers_pt > feat_th * 4)  // unreachable upstream too (shadowed by the branch above)
            orb_fast_th = std::min(max_fast, orb_fast_th + 2 * fast_inc);
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
    for (size_t i1 = 0; i1 < matches_12.size(); ++i1) {
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
    for (size_t i1 = 0; i1 < matches_12.size(); ++i1) {
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
    t_pose_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_begin).count();
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
