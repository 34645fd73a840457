// pm_host.cpp — TEST INFRASTRUCTURE ONLY: exposes the product's pose_math.h (the FP64 building
// blocks the HIP pose kernel is made of) to the CPU test-suite, so that the algebra the GPU runs
// is checked against the oracle in this GPU-less container.  Not part of the product library.
#include "../../stvo-pl_amd/csrc/pose_math.h"
#include "../../stvo-pl_amd/host/keyframe.h"

extern "C" {
double pmh_det6(const double* A) { return pm::det6(A); }
void pmh_unctinv(const double* T, const double* cov, double* out) { pm::uncTinv_se3(T, cov, out); }
// key-frame decision over a sequence of n frames: inputs Tfw[n][16], DT[n][16], DTcov[n][36]; a frame that needs a new
// key-frame is made one (its Tfw becomes identity for the state, like currFrameIsKF).  out[n] = decision, acc[n][36] =
// accumulated covariance after each call, ent[n] = entropy_first_prevKF after each call.
void pmh_kf_sequence(int n, const double* Tfw, const double* DT, const double* DTcov, double min_ratio, double max_t, double max_r,
                     int* out, double* acc, double* ent) {
    StVO::KeyFrameState k;
    for (int f = 0; f < n; ++f) {
        StVO::Matrix4d T, D;
        StVO::Matrix6d C;
        for (int i = 0; i < 16; ++i) { T.m[i] = Tfw[f * 16 + i]; D.m[i] = DT[f * 16 + i]; }
        for (int i = 0; i < 36; ++i) C.m[i] = DTcov[f * 36 + i];
        out[f] = StVO::kf_need_new(k, T, D, C, min_ratio, max_t, max_r, false) ? 1 : 0;
        for (int i = 0; i < 36; ++i) acc[f * 36 + i] = k.cov_prevKF_currF.m[i];
        ent[f] = k.entropy_first_prevKF;
        if (out[f]) StVO::kf_reset(k, StVO::Matrix4d::Identity());
    }
}
void pmh_expmap(const double* x, double* T) { pm::expmap_se3(x, T); }
void pmh_logmap(const double* T, double* x) { pm::logmap_se3(T, x); }
void pmh_inverse_se3(const double* T, double* Ti) { pm::inverse_se3(T, Ti); }
void pmh_adjoint(const double* T, double* A) { pm::adjoint_se3(T, A); }
void pmh_unccomp(const double* T1, const double* c1, const double* ci, double* out) { pm::unccomp_se3(T1, c1, ci, out); }
int pmh_solve6(const double* H, const double* g, double* x, double* lad) { return pm::solve6(H, g, x, lad); }
void pmh_inverse6(const double* A, double* Ai) { pm::inverse6(A, Ai); }
// the memory-operand forms of the two pivoted routines (round 6): same operations, same order
int pmh_solve6_mem(const double* H, const double* g, double* x, double* lad) {
    double A[36], c[6], y[6];
    int perm[6];
    for (int i = 0; i < 36; ++i) A[i] = H[i];
    for (int i = 0; i < 6; ++i) c[i] = g[i];
    const int rank = pm::solve6_mem(A, c, y, perm, c, lad);   // (the solution aliases the right-hand side, as on the device)
    for (int i = 0; i < 6; ++i) x[i] = c[i];
    return rank;
}
void pmh_inverse6_mem(const double* Ain, double* Ai) {
    double A[36];
    for (int i = 0; i < 36; ++i) A[i] = Ain[i];
    pm::inverse6_mem(A, Ai);
}
void pmh_eig6(const double* A, double* w) { pm::eig6(A, w); }
void pmh_eig6_ql(const double* A, double* w) { pm::eig6_ql(A, w); }
int pmh_solve6_spd(const double* H, const double* g, double* x, double* lad) { return pm::solve6_spd(H, g, x, lad) ? 1 : 0; }
int pmh_inverse6_spd(const double* A, double* Ai) { return pm::inverse6_spd(A, Ai) ? 1 : 0; }
void pmh_step_pose(double* DT, const double* inc) { pm::step_pose(DT, inc); }
int pmh_spd_cert(const double* C) { return pm::spd_unit_certificate(C); }
double pmh_line_overlap(const double* so, const double* eo, const double* sp, const double* ep) {
    return pm::line_overlap(so[0], so[1], eo[0], eo[1], sp[0], sp[1], ep[0], ep[1]);
}
// serial emulation of one optimizeFunctions evaluation using the device per-feature terms
void pmh_normal_eq(const double* DT, const stvo_cam* cam, double homog_th, const stvo_matched* m, int robust, double s_p,
                   double s_l, double* acc28) {
    pm::Cam5 c{cam->fx, cam->fy, cam->cx, cam->cy};
    for (int i = 0; i < 28; ++i) acc28[i] = 0.0;
    for (int i = 0; i < m->np; ++i)
        if (m->inlier_p[i])
            pm::point_term(acc28, DT, c, homog_th, m->P[3 * i], m->P[3 * i + 1], m->P[3 * i + 2], m->pl_obs[2 * i],
                           m->pl_obs[2 * i + 1], m->sigma2p[i], robust != 0, s_p);
    for (int i = 0; i < m->nl; ++i)
        if (m->inlier_l[i]) {
            pm::LineRec L;
            for (int k = 0; k < 3; ++k) {
                L.sP[k] = m->sP[3 * i + k];
                L.eP[k] = m->eP[3 * i + k];
                L.le[k] = m->le_obs[3 * i + k];
            }
            for (int k = 0; k < 2; ++k) {
                L.spl[k] = m->spl[2 * i + k];
                L.epl[k] = m->epl[2 * i + k];
            }
            L.sigma2 = m->sigma2l[i];
            pm::line_term(acc28, DT, c, homog_th, L, robust != 0, s_l);
        }
}
}
