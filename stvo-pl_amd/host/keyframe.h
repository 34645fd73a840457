// keyframe.h — the key-frame decision of StVO::StereoFrameHandler (the "slam functions" of
// /root/reference/src/stereoFrameHandler.cpp:1134-1218, used by PL-SLAM on top of PL-StVO), as plain host
// code with no GPU dependency: state (include/stereoFrameHandler.h:81-85) + needNewKF + currFrameIsKF.
// SURVEY.md §8f rank 2.  The handler mirror forwards to these; tests/test_pose_math_host.py checks them
// against a numpy model without a GPU.
#pragma once

#include <cmath>
#include <iostream>

#include "../csrc/pose_math.h"
#include "stvo_compat.h"

namespace StVO {

struct KeyFrameState {
    bool prev_f_iskf = true;                       // stereoFrameHandler.cpp:50
    double entropy_first_prevKF = 0.0;
    Matrix4d T_prevKF = Matrix4d::Identity();      // :48
    Matrix6d cov_prevKF_currF = Matrix6d::Zero();  // :49
    int N_prevKF_currF = 0;                        // :51
};

// needNewKF (:1136-1188).  Tfw, DT, DT_cov are curr_frame's fields.  Returns true when a new key-frame is needed;
// otherwise counts the frame (N_prevKF_currF++).  The accumulated covariance is updated in both cases, like the original.
inline bool kf_need_new(KeyFrameState& k, const Matrix4d& Tfw, const Matrix4d& DT, const Matrix6d& DT_cov,
                        double min_entropy_ratio, double max_kf_t_dist, double max_kf_r_dist, bool verbose = true) {
    const double kPi = 3.1415926535897932384626433832795;  // CV_PI
    const double two_pi_term = 3.0 * (1.0 + std::log(2.0 * std::acos(-1.0)));
    if (k.prev_f_iskf) {  // :1140-1153 — first frame after a key-frame fixes the reference entropy
        const double det = pm::det6(DT_cov.m);
        k.entropy_first_prevKF = (det != 0.0) ? two_pi_term + 0.5 * std::log(det) : -999999999.99;
        k.prev_f_iskf = false;
    }
    // geometric distance from the previous key-frame (:1156-1159)
    double Ti[16], D[16], dX[6];
    pm::inverse_se3(Tfw.m, Ti);
    pm::mat4_mul(Ti, k.T_prevKF.m, D);
    pm::logmap_se3(D, dX);
    const double t = std::sqrt(dX[0] * dX[0] + dX[1] * dX[1] + dX[2] * dX[2]);
    const double r = std::sqrt(dX[3] * dX[3] + dX[4] * dX[4] + dX[5] * dX[5]) * 180.f / kPi;
    // accumulated covariance from the previous key-frame (:1162-1166)
    double A[36], cinv[36], tmp[36];
    pm::adjoint_se3(k.T_prevKF.m, A);
    pm::uncTinv_se3(DT.m, DT_cov.m, cinv);
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            for (int q = 0; q < 6; ++q) s += A[i * 6 + q] * cinv[q * 6 + j];
            tmp[i * 6 + j] = s;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            for (int q = 0; q < 6; ++q) s += tmp[i * 6 + q] * A[j * 6 + q];
            k.cov_prevKF_currF.m[i * 6 + j] += s;
        }
    const double det_acc = pm::det6(k.cov_prevKF_currF.m);
    const double entropy_curr = two_pi_term + 0.5 * std::log(det_acc);
    const double entropy_ratio = entropy_curr / k.entropy_first_prevKF;
    bool zero_cov = true, ident = true;
    for (int i = 0; i < 36; ++i) zero_cov = zero_cov && DT_cov.m[i] == 0.0;
    for (int i = 0; i < 16; ++i) ident = ident && DT.m[i] == ((i % 5 == 0) ? 1.0 : 0.0);
    // :1173-1175
    if (entropy_ratio < min_entropy_ratio || std::isnan(entropy_ratio) || std::isinf(entropy_ratio) || (zero_cov && ident) ||
        t > max_kf_t_dist || r > max_kf_r_dist || k.N_prevKF_currF > 10) {
        if (verbose)
            std::cout << std::endl << "Entropy ratio: " << entropy_ratio << "\t" << t << " " << r << " " << k.N_prevKF_currF << std::endl;
        return true;
    }
    if (verbose)
        std::cout << std::endl << "No new KF needed: " << entropy_ratio << "\t" << entropy_curr << " " << k.entropy_first_prevKF << " "
                  << det_acc << "\t" << t << " " << r << " " << k.N_prevKF_currF << std::endl << std::endl;
    k.N_prevKF_currF++;
    return false;
}

// the state part of currFrameIsKF (:1209-1216); the caller resets the feature indices and the frame pose
inline void kf_reset(KeyFrameState& k, const Matrix4d& Tfw_of_new_kf) {
    k.T_prevKF = Tfw_of_new_kf;
    k.cov_prevKF_currF = Matrix6d::Zero();
    k.prev_f_iskf = true;
    k.N_prevKF_currF = 0;
}

}  // namespace StVO
