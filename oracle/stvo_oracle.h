/*
 * stvo_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99, scalar, single-threaded) of the reference's per-frame hot path:
 * binary-descriptor matching and StereoFrameHandler::optimizePose.  It exists to CHECK the HIP
 * product path; nothing in the product (stvo-pl_amd/, include/) may include, link or call it.
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it.
 *
 * PINNING STATUS:
 *   - grid bucketing + Bresenham rasteriser (orc_line_coords, orc_grid_window_gather): PINNED against
 *     the reference's own src/gridStructure.cpp + src/lineIterator.cpp compiled unmodified into
 *     oracle/_ref/libstvo_ref.so (tests/test_oracle_ref.py).
 *   - matcher (src/matching.cpp) and optimizer (src/stereoFrameHandler.cpp): PARITY UNPINNED.
 *     The reference has no tests / golden vectors and those translation units cannot
 *     be built here without stand-ins for OpenCV 3.x and Eigen 3 (absent from the image and from
 *     /root/reference).  They are restated from the cited lines and cross-checked against an
 *     independent numpy float64 model (tests/np_model.py) and analytic invariants only.
 *   - third-party arithmetic restated from its published algorithm, not from source:
 *     cv::BFMatcher::knnMatch (OpenCV 3.x, call site src/matching.cpp:47-48) and Eigen 3
 *     ColPivHouseholderQR / PartialPivLU inverse / SelfAdjointEigenSolver (call sites
 *     src/stereoFrameHandler.cpp:294,417-418,429).
 */
#ifndef STVO_ORACLE_H
#define STVO_ORACLE_H

#include "../include/stvo_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- matching half ------------------------------------------------------------------------ */
int orc_distance(const uint8_t* a, const uint8_t* b);
void orc_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx0, int32_t* d0, int32_t* d1);
int orc_match_nnr(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, int32_t* m12);
int orc_match(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, int best_lr, int32_t* m12);

/* list-grid restated as CSR: cell index = x * rows + y is NOT used; we use y * cols + x. */
void orc_grid_build(const int32_t* cell_xy, const int32_t* owner, int n_entries, int32_t* cell_start /*[cells+1]*/,
                    int32_t* cell_items /*[<=n_entries]*/);
int orc_line_coords(double x1, double y1, double x2, double y2, int32_t* out_xy, int max_cells);
int orc_grid_window_gather(const int32_t* cell_start, const int32_t* cell_items, int x, int y, const stvo_grid_window* w,
                           int32_t* out, uint8_t* seen /*[n2], zeroed; restored on return*/);

int orc_match_grid_points(const int32_t* cell_xy1, const uint8_t* d1, int n1, const int32_t* cell_start,
                          const int32_t* cell_items, const uint8_t* d2, int n2, const stvo_grid_window* w, double ratio,
                          int best_lr, int32_t* m12);
int orc_match_grid_lines(const int32_t* cell_xy1 /*[n1][4] sx,sy,ex,ey*/, const uint8_t* d1, int n1,
                         const int32_t* cell_start, const int32_t* cell_items, const uint8_t* d2, int n2,
                         const double* dir2 /*[n2][2]*/, const stvo_grid_window* w, double ratio, double line_sim_th,
                         int best_lr, int32_t* m12);

/* stereo association glue (StereoFrame::matchStereoPoints / matchStereoLines) */
int orc_stereo_points(const float* kp_l /*[n][2]*/, const int32_t* oct_l, const uint8_t* desc_l, int nl,
                      const float* kp_r, const uint8_t* desc_r, int nr, int img_cols, int img_rows, const stvo_cam* cam,
                      const stvo_match_params* mp,
                      /* out, capacity nl: */ int32_t* src_idx, double* pl /*[k][2]*/, double* disp, double* P /*[k][3]*/,
                      double* sigma2, int32_t* m12_raw /*[nl] or NULL*/);
int orc_stereo_lines(const float* kl_l /*[n][4] sx,sy,ex,ey*/, const float* angle_l, const int32_t* oct_l,
                     const uint8_t* desc_l, int nl, const float* kl_r, const uint8_t* desc_r, int nr, int img_cols,
                     int img_rows, const stvo_cam* cam, const stvo_match_params* mp,
                     /* out, capacity nl: */ int32_t* src_idx, double* spl, double* epl, double* sdisp, double* edisp,
                     double* sP, double* eP, double* le, double* sigma2, int32_t* m12_raw);

/* ---- optimizer half ----------------------------------------------------------------------- */
void orc_expmap_se3(const double x[6], double T[16]);
void orc_logmap_se3(const double T[16], double x[6]);
void orc_inverse_se3(const double T[16], double Ti[16]);
void orc_adjoint_se3(const double T[16], double A[36]);
void orc_unccomp_se3(const double T1[16], const double cov1[36], const double covinc[36], double out[36]);
int orc_solve6(const double H[36], const double g[6], double x[6], double* log_abs_det);
void orc_inverse6(const double A[36], double Ai[36]);
void orc_eig6(const double A[36], double w[6]);
void orc_mean_stdv_mad(const double* r, int n, double* mean, double* stdv);
double orc_stdv_mad(const double* r, int n);
double orc_line_overlap(const double spl_obs[2], const double epl_obs[2], const double spl_proj[2],
                        const double epl_proj[2]);
double orc_line_overlap_stereo(double spl_obs, double epl_obs, double spl_proj, double epl_proj, double line_horiz_th);
int orc_is_good_solution(const double DT[16], const double cov[36], double err);

/* correspondence records (matched_pt / matched_ls), arrays of structs-of-arrays */
typedef struct orc_matched {
    int np;
    const double* P;       /* [np][3]  PointFeature::P                                  */
    const double* pl_obs;  /* [np][2]  PointFeature::pl_obs                             */
    const double* sigma2p; /* [np]                                                      */
    int32_t* inlier_p;     /* [np] in/out                                               */
    int nl;
    const double* sP;      /* [nl][3] */
    const double* eP;      /* [nl][3] */
    const double* le_obs;  /* [nl][3] */
    const double* spl;     /* [nl][2]  prev-frame endpoints used by the overlap weight  */
    const double* epl;     /* [nl][2] */
    const double* sigma2l; /* [nl] */
    int32_t* inlier_l;     /* [nl] in/out */
} orc_matched;

void orc_optimize_functions(const double DT[16], const stvo_cam* cam, const stvo_opt_params* p, const orc_matched* m,
                            int robust, double H[36], double g[6], double* e, int32_t* n_used);
void orc_remove_outliers(const double DT[16], const stvo_cam* cam, const stvo_opt_params* p, orc_matched* m,
                         int32_t* n_inl_pt, int32_t* n_inl_ls);
int orc_need_new_kf(double* state55, const double* Tfw, const double* DT, const double* DT_cov, double min_entropy_ratio,
                    double max_kf_t_dist, double max_kf_r_dist);
void orc_curr_frame_is_kf(double* state55);
void orc_optimize_pose(const double init_T[16], const stvo_cam* cam, const stvo_opt_params* p, orc_matched* m,
                       stvo_pose_result* out);

#ifdef __cplusplus
}
#endif
#endif
