// imagesStVO_synth.cpp — the caller of the boundary: the per-frame loop of
// /root/reference/app/imagesStVO.cpp:86-124 (initialize / insertStereoPair / optimizePose /
// updateFrame, the "Proc. time" timer and the console line of :114-121), driven by a file of
// pre-extracted stereo features instead of images (no OpenCV / datasets in this image; the
// ORB / LSD+LBD front-end is out of scope).  Writes per-frame results for the parity tests.
//
//   imagesStVO_synth <sequence.bin> <results.bin> [--preset kitti|euroc|default] [-c config.yaml] [--mode 0|1|2]
//                    [-o offset] [-n N] [-s step]   (the reference's options, app/imagesStVO.cpp:138-171)
//                    [--keyframes]   (needNewKF / currFrameIsKF after every optimizePose, as PL-SLAM drives them)
//                    [--device-pipeline]   (stvo_seq_*: one upload + one synchronisation per frame, state in HBM)
//                    [--no-lines]    (Config::hasLines() = false)
// A sequence file that starts with "STVOIMG1" holds stereo IMAGES (n_frames, cols, rows, camera, then per frame the left and the
// right 8-bit image): the loop then calls the reference's own entry points initialize / insertStereoPair(img_l, img_r, idx)
// (include/stereoFrameHandler.h:44-45), whose ORB point front-end runs on the GPU (key-points only: no LSD / LBD here).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../host/stereoFrameHandler.h"

using namespace StVO;

namespace {
template <typename T>
bool rd(std::ifstream& f, T* p, size_t n = 1) {
    f.read(reinterpret_cast<char*>(p), (std::streamsize)(sizeof(T) * n));
    return (bool)f;
}
template <typename T>
void wr(std::ofstream& f, const T* p, size_t n = 1) {
    f.write(reinterpret_cast<const char*>(p), (std::streamsize)(sizeof(T) * n));
}

bool read_points(std::ifstream& f, int n, std::vector<KeyPoint>& kp, DescMat& d) {
    kp.resize(n);
    for (int i = 0; i < n; ++i) {
        if (!rd(f, &kp[i].x) || !rd(f, &kp[i].y) || !rd(f, &kp[i].octave)) return false;
    }
    d.rows = n;
    d.data.resize((size_t)n * 32);
    return n == 0 || rd(f, d.data.data(), (size_t)n * 32);
}
bool read_lines(std::ifstream& f, int n, std::vector<KeyLine>& kl, DescMat& d) {
    kl.resize(n);
    for (int i = 0; i < n; ++i) {
        if (!rd(f, &kl[i].startPointX) || !rd(f, &kl[i].startPointY) || !rd(f, &kl[i].endPointX) ||
            !rd(f, &kl[i].endPointY) || !rd(f, &kl[i].angle) || !rd(f, &kl[i].octave))
            return false;
    }
    d.rows = n;
    d.data.resize((size_t)n * 32);
    return n == 0 || rd(f, d.data.data(), (size_t)n * 32);
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        std::cerr << "usage: imagesStVO_synth <sequence.bin> <results.bin> [--preset kitti|euroc|default] [-c cfg] "
                     "[--mode m] [-n frames]\n";
        return -1;
    }
    std::string preset = "kitti", cfg;
    int mode = 0, max_frames = 0, frame_offset = 0, frame_step = 1;
    bool keyframes = false;
    bool device_pipeline = false, no_lines = false;
    for (int i = 3; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--preset") && i + 1 < argc) preset = argv[++i];
        else if (!std::strcmp(argv[i], "-c") && i + 1 < argc) cfg = argv[++i];
        else if (!std::strcmp(argv[i], "--mode") && i + 1 < argc) mode = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "-n") && i + 1 < argc) max_frames = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--device-pipeline")) device_pipeline = true;
        else if (!std::strcmp(argv[i], "--keyframes")) keyframes = true;
        else if (!std::strcmp(argv[i], "--no-lines")) no_lines = true;
        else if (!std::strcmp(argv[i], "-o") && i + 1 < argc) frame_offset = std::atoi(argv[++i]);  // imagesStVO.cpp:158-159
        else if (!std::strcmp(argv[i], "-s") && i + 1 < argc) frame_step = std::atoi(argv[++i]);    // imagesStVO.cpp:162-163
    }
    if (preset == "kitti") Config::setKittiPreset();
    else if (preset == "euroc") Config::setEurocPreset();
    else Config::setDefaults();
    if (!cfg.empty()) Config::loadFromFile(cfg);
    if (no_lines) Config::hasLines() = false;

    std::ifstream in(argv[1], std::ios::binary);
    char magic[8];
    int32_t n_frames = 0, cols = 0, rows = 0;
    double camv[5];
    if (!in || !rd(in, magic, 8) || (std::memcmp(magic, "STVOSEQ1", 8) != 0 && std::memcmp(magic, "STVOIMG1", 8) != 0) || !rd(in, &n_frames) ||
        !rd(in, &cols) || !rd(in, &rows) || !rd(in, camv, 5)) {
        std::cerr << "bad sequence file\n";
        return -1;
    }
    const bool image_file = std::memcmp(magic, "STVOIMG1", 8) == 0;
    if (image_file && device_pipeline) {
        std::cerr << "--device-pipeline takes feature files; image files go through the handler" << std::endl;
        return -1;
    }
    if (frame_offset < 0) frame_offset = 0;
    if (frame_step < 1) frame_step = 1;
    const int n_file_frames = n_frames;
    {   // frames the run will process: offset, step, then -n
        const int avail = n_file_frames > frame_offset ? (n_file_frames - frame_offset + frame_step - 1) / frame_step : 0;
        n_frames = (max_frames > 0 && max_frames < avail) ? max_frames : avail;
    }
    std::ofstream out(argv[2], std::ios::binary);
    PinholeStereoCamera* cam_pin = new PinholeStereoCamera(cols, rows, camv[0], camv[1], camv[2], camv[3], camv[4]);

    if (device_pipeline) {
        // ---- the same loop on the device-resident pipeline: pose + counters per frame, Tfw composed here
        stvo_ctx* ctx = nullptr;
        if (stvo_ctx_create(0, 2048, 1, &ctx) != STVO_OK) {
            std::cerr << "[StVO-HIP] no MI355X available" << std::endl;
            return -2;
        }
        stvo_match_params mp{};
        mp.best_lr_matches = Config::bestLRMatches(); mp.matching_s_ws = Config::matchingSWs();
        mp.min_ratio_12_p = (float)Config::minRatio12P(); mp.min_ratio_12_l = (float)Config::minRatio12L();
        mp.max_dist_epip = Config::maxDistEpip(); mp.min_disp = Config::minDisp(); mp.line_sim_th = Config::lineSimTh();
        mp.stereo_overlap_th = Config::stereoOverlapTh(); mp.line_horiz_th = Config::lineHorizTh();
        mp.ls_min_disp_ratio = Config::lsMinDispRatio(); mp.orb_scale_factor = Config::orbScaleFactor();
        mp.lsd_scale = Config::lsdScale();
        mp.min_ratio_12_p_d = Config::minRatio12P();
        if (frame_offset != 0 || frame_step != 1 || keyframes) {
            std::cerr << "--device-pipeline takes the sequence as it is: -o / -s / --keyframes belong to the handler path" << std::endl;
            return -1;
        }
        stvo_opt_params op{};
        op.mode = mode; op.has_points = Config::hasPoints(); op.has_lines = Config::hasLines();
        op.min_features = Config::minFeatures(); op.max_iters = Config::maxIters(); op.max_iters_ref = Config::maxItersRef();
        op.homog_th = Config::homogTh(); op.min_error = Config::minError(); op.min_error_change = Config::minErrorChange();
        op.inlier_k = Config::inlierK();
        const stvo_cam cam = cam_pin->abi();
        stvo_seq* seq = nullptr;
        if (stvo_seq_create(ctx, 1, 2048, 512, cols, rows, &cam, &mp, &op, &seq) != STVO_OK) {
            std::cerr << "stvo_seq_create failed" << std::endl;
            return -2;
        }
        double t_sum = 0.0;
        std::vector<double> t_all;
        for (int k = 0; k < n_frames; ++k) {
            FrameFeatures feat;
            int32_t n[4];
            if (!rd(in, n, 4) || !read_points(in, n[0], feat.points_l, feat.pdesc_l) ||
                !read_points(in, n[1], feat.points_r, feat.pdesc_r) || !read_lines(in, n[2], feat.lines_l, feat.ldesc_l) ||
                !read_lines(in, n[3], feat.lines_r, feat.ldesc_r))
                return -1;
            std::vector<float> kpl(2 * n[0] + 2), kpr(2 * n[1] + 2), kll(4 * n[2] + 4), klr(4 * n[3] + 4);
            std::vector<int32_t> ol(n[0] + 1), oll(n[2] + 1);
            for (int i = 0; i < n[0]; ++i) { kpl[2 * i] = feat.points_l[i].x; kpl[2 * i + 1] = feat.points_l[i].y; ol[i] = feat.points_l[i].octave; }
            for (int i = 0; i < n[1]; ++i) { kpr[2 * i] = feat.points_r[i].x; kpr[2 * i + 1] = feat.points_r[i].y; }
            for (int i = 0; i < n[2]; ++i) { kll[4 * i] = feat.lines_l[i].startPointX; kll[4 * i + 1] = feat.lines_l[i].startPointY;
                                             kll[4 * i + 2] = feat.lines_l[i].endPointX; kll[4 * i + 3] = feat.lines_l[i].endPointY; oll[i] = feat.lines_l[i].octave; }
            for (int i = 0; i < n[3]; ++i) { klr[4 * i] = feat.lines_r[i].startPointX; klr[4 * i + 1] = feat.lines_r[i].startPointY;
                                             klr[4 * i + 2] = feat.lines_r[i].endPointX; klr[4 * i + 3] = feat.lines_r[i].endPointY; }
            stvo_frame_features ff{};
            ff.stride_kp = n[0] > n[1] ? n[0] : n[1];
            ff.stride_kl = n[2] > n[3] ? n[2] : n[3];
            ff.n_kp_l = &n[0]; ff.n_kp_r = &n[1]; ff.n_kl_l = &n[2]; ff.n_kl_r = &n[3];
            ff.kp_l = kpl.data(); ff.oct_l = ol.data(); ff.desc_l = feat.pdesc_l.ptr(); ff.kp_r = kpr.data(); ff.desc_r = feat.pdesc_r.ptr();
            ff.kl_l = kll.data(); ff.oct_ll = oll.data(); ff.ldesc_l = feat.ldesc_l.ptr(); ff.kl_r = klr.data(); ff.ldesc_r = feat.ldesc_r.ptr();
            stvo_pose_result r{};
            int32_t cnt[4];
            const auto t0 = std::chrono::high_resolution_clock::now();
            if (stvo_seq_push(seq, &ff, &r, cnt) != STVO_OK) {
                std::cerr << "stvo_seq_push failed: " << stvo_ctx_last_error(ctx) << std::endl;
                return -3;
            }
            const double t1 = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
            if (k == 0) continue;
            t_sum += t1;
            t_all.push_back(t1);
            std::printf("Frame: %d\tRes.: %.8f \t Proc. time: %.3f ms\t \t Points: %d (%d) \t Lines:  %d (%d) \n", k, r.err, t1,
                        r.n_matched_pt, r.n_inliers_pt, r.n_matched_ls, r.n_inliers_ls);
            const int32_t ints[12] = {k, r.status, r.path, r.iters[0], r.iters[1], r.n_matched_pt, r.n_inliers_pt, r.n_matched_ls,
                                      r.n_inliers_ls, cnt[0], cnt[1], 0};
            wr(out, ints, 12);
            wr(out, r.T, 16);
            wr(out, r.cov, 36);
            wr(out, r.cov_eig, 6);
            wr(out, &r.err, 1);
            const double zeros[52] = {0};
            wr(out, zeros, 52);  // Tfw / Tfw_cov are composed by the caller in this mode
            const int32_t z2[2] = {0, 0};
            wr(out, z2, 2);
        }
        if (n_frames > 1) {
            // the median is the steady-state figure: the first frames of a process pay for code-object loading and allocations
            std::sort(t_all.begin(), t_all.end());
            std::printf("[imagesStVO_synth --device-pipeline] %d frame pairs, mean Proc. time %.3f ms, median %.3f ms (single stream, incl. H2D/D2H)\n",
                        n_frames - 1, t_sum / (n_frames - 1), t_all[t_all.size() / 2]);
        }
        stvo_seq_destroy(seq);
        stvo_ctx_destroy(ctx);
        delete cam_pin;
        return 0;
    }

    StereoFrameHandler* StVO = nullptr;
    try {
        StVO = new StereoFrameHandler(cam_pin);
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        return -2;
    }
    StVO->mode = mode;
    double t_total = 0.0, t_st = 0.0, t_ff = 0.0, t_po = 0.0;
    std::vector<double> t_all;
    // -o / -s / -n as the reference's Dataset applies them (src/dataset.cpp: skip `offset` frames, then take every `step`-th
    // frame, at most `n` of them): frames that are not selected are read and dropped
    int frame_counter = -1, n_done = 0;
    for (int file_idx = 0; file_idx < n_file_frames; ++file_idx) {
        FrameFeatures feat;
        feat.img_cols = cols;
        feat.img_rows = rows;
        std::vector<uint8_t> img_l, img_r;
        if (image_file) {
            const size_t px = (size_t)cols * rows;
            img_l.resize(px);
            img_r.resize(px);
            if (!rd(in, img_l.data(), px) || !rd(in, img_r.data(), px)) {
                std::cerr << "truncated image file at frame " << file_idx << "\n";
                return -1;
            }
        } else {
            int32_t n[4];
            if (!rd(in, n, 4) || !read_points(in, n[0], feat.points_l, feat.pdesc_l) ||
                !read_points(in, n[1], feat.points_r, feat.pdesc_r) || !read_lines(in, n[2], feat.lines_l, feat.ldesc_l) ||
                !read_lines(in, n[3], feat.lines_r, feat.ldesc_r)) {
                std::cerr << "truncated sequence file at frame " << file_idx << "\n";
                return -1;
            }
        }
        if (file_idx < frame_offset || (file_idx - frame_offset) % frame_step != 0) continue;
        if (max_frames > 0 && n_done >= max_frames) break;
        ++frame_counter;
        ++n_done;
        const GrayImage gl{img_l.data(), rows, cols, (size_t)cols}, gr{img_r.data(), rows, cols, (size_t)cols};
        if (frame_counter == 0) {
            if (image_file) StVO->initialize(gl, gr, 0);  // initialize(img_l, img_r, 0), imagesStVO.cpp:90
            else StVO->initialize(feat, 0);
            continue;
        }
        const auto t0 = std::chrono::high_resolution_clock::now();  // timer.start()  (imagesStVO.cpp:95)
        if (image_file) StVO->insertStereoPair(gl, gr, frame_counter);  // imagesStVO.cpp:96
        else StVO->insertStereoPair(feat, frame_counter);
        StVO->optimizePose();
        const double t1 =
            std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
        t_total += t1;
        t_all.push_back(t1);
        t_st += StVO->t_stereo_ms;
        t_ff += StVO->t_f2f_ms;
        t_po += StVO->t_pose_ms;

        // console output (imagesStVO.cpp:114-121)
        std::printf("Frame: %d\tRes.: %.8f \t Proc. time: %.3f ms\t ", frame_counter, StVO->curr_frame->err_norm, t1);
        if (Config::adaptativeFAST()) std::printf("\t FAST: %d", StVO->orb_fast_th);
        if (Config::hasPoints()) std::printf("\t Points: %zu (%d) ", StVO->matched_pt.size(), StVO->n_inliers_pt);
        if (Config::hasLines()) std::printf("\t Lines:  %zu (%d) ", StVO->matched_ls.size(), StVO->n_inliers_ls);
        std::printf("\n");

        const stvo_pose_result& r = StVO->last_result;
        const int32_t ints[12] = {frame_counter, r.status, r.path, r.iters[0], r.iters[1], (int32_t)StVO->matched_pt.size(),
                                  StVO->n_inliers_pt, (int32_t)StVO->matched_ls.size(), StVO->n_inliers_ls,
                                  (int32_t)StVO->curr_frame->stereo_pt.size(), (int32_t)StVO->curr_frame->stereo_ls.size(),
                                  0};
        int32_t new_kf = 0;
        wr(out, ints, 12);
        wr(out, StVO->curr_frame->DT.m, 16);
        wr(out, StVO->curr_frame->DT_cov.m, 36);
        wr(out, StVO->curr_frame->DT_cov_eig.v, 6);
        wr(out, &StVO->curr_frame->err_norm, 1);
        wr(out, StVO->curr_frame->Tfw.m, 16);
        wr(out, StVO->curr_frame->Tfw_cov.m, 36);
        // the key-frame decision PL-SLAM drives on top of the odometry (src/stereoFrameHandler.cpp:1136-1218): the pose of the
        // frame has been written above; a new key-frame restarts the map frame (Tfw = I) for the frames that follow
        if (keyframes && StVO->needNewKF()) {
            StVO->currFrameIsKF();
            new_kf = 1;
        }
        StVO->updateFrame();
        const int32_t fast = StVO->orb_fast_th;
        wr(out, &fast, 1);
        wr(out, &new_kf, 1);  // (the spare word of the record)
    }
    if (n_frames > 1 && !t_all.empty()) {
        std::sort(t_all.begin(), t_all.end());
        std::printf("[imagesStVO_synth] %d frame pairs, mean Proc. time %.3f ms, median %.3f ms (single stream, incl. H2D/D2H): stereo "
                    "association %.3f, f2f matching %.3f, optimizePose %.3f\n",
                    n_frames - 1, t_total / (n_frames - 1), t_all[t_all.size() / 2], t_st / (n_frames - 1), t_ff / (n_frames - 1),
                    t_po / (n_frames - 1));
    }
    delete StVO;
    delete cam_pin;
    return 0;
}
