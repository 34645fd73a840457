/*
 * stvo_types.h — plain-C data records shared by the C-ABI (stvo_hip.h), the host-side
 * mirror of the reference classes, and the test oracle.  POD only: no torch, no C++.
 *
 * Conventions (all follow the reference, /root/reference):
 *   - 4x4 / 6x6 matrices are row-major doubles.
 *   - se(3) twists are ordered (t, w): head(3) translation, tail(3) rotation
 *     (src/auxiliar.cpp:124-141).
 *   - Binary descriptors are N x 32 bytes, contiguous (ORB 256 bit, LBD 256 bit;
 *     src/stereoFrame.cpp:112-115, 3rdparty/line_descriptor/src/binary_descriptor_custom.cpp:639).
 *   - The stereo bucketing grid is 64 columns x 48 rows (include/stereoFrame.h:51-52).
 */
#ifndef STVO_TYPES_H
#define STVO_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STVO_DESC_BYTES 32
#define STVO_GRID_COLS 64 /* include/stereoFrame.h:52 */
#define STVO_GRID_ROWS 48 /* include/stereoFrame.h:51 */
#define STVO_GRID_CELLS (STVO_GRID_COLS * STVO_GRID_ROWS)

/* Pinhole stereo camera scalars read by the path (src/pinholeStereoCamera.cpp:221-237). */
typedef struct stvo_cam {
    double fx, fy, cx, cy, b;
} stvo_cam;

/* GridWindow (include/gridStructure.h:37-39): cells [x-w_lo .. x+w_hi] x [y-h_lo .. y+h_hi]. */
typedef struct stvo_grid_window {
    int32_t w_lo, w_hi, h_lo, h_hi;
} stvo_grid_window;

/* Optimizer parameters = the Config scalars optimizePose reads (src/config.cpp:80-86,
 * include/config.h:39-105).  `mode` is the local constant of stereoFrameHandler.cpp:329
 * (0 GN, 1 robust GN, 2 LM) exposed as a parameter; the reference hard-codes 0. */
typedef struct stvo_opt_params {
    int32_t mode;
    int32_t has_points;
    int32_t has_lines;
    int32_t min_features;
    int32_t max_iters;
    int32_t max_iters_ref;
    int32_t reserved0;
    int32_t reserved1;
    double homog_th;
    double min_error;
    double min_error_change;
    double inlier_k;
} stvo_opt_params;

/* Matching / stereo-association parameters (src/config.cpp:49-69,91). */
typedef struct stvo_match_params {
    int32_t best_lr_matches;
    int32_t matching_s_ws;
    float   min_ratio_12_p; /* passed as float to match() (src/matching.cpp:63) */
    float   min_ratio_12_l;
    double  max_dist_epip;
    double  min_disp;
    double  line_sim_th;
    double  stereo_overlap_th;
    double  line_horiz_th;
    double  ls_min_disp_ratio;
    double  orb_scale_factor;
    double  lsd_scale;
    /* Config::minRatio12P() as the DOUBLE both matchGrid overloads compare with (src/matching.cpp:160,241), used by the
     * stereo stage of the stvo_seq_* pipeline; 0 = (double)min_ratio_12_p.  (match() takes the ratio as float,
     * src/matching.cpp:63, so the f2f stage keeps the float fields above.) */
    double  min_ratio_12_p_d;
} stvo_match_params;

/* Status of one optimizePose call (the in-band failure modes of
 * src/stereoFrameHandler.cpp:332-391). */
enum {
    STVO_POSE_OK = 0,                  /* committed: good solution and DT != I           (:372-381) */
    STVO_POSE_FEW_INLIERS_BEFORE = 1,  /* n_inliers < minFeatures before optimisation    (:364-368) */
    STVO_POSE_FEW_INLIERS_AFTER = 2,   /* n_inliers < minFeatures after removeOutliers   (:351-355) */
    STVO_POSE_REJECTED = 3,            /* isGoodSolution false or DT == I at commit      (:382-391) */
    STVO_POSE_INTERNAL = 4             /* device-side failure: the single-stream pose kernel waited ~2 s for the key-line stream's
                                          signal and did not get it (a failed launch there); pose held, no by-products published */
};

/* Flags describing the path taken through the state machine. */
enum {
    STVO_PATH_STAGE1_GOOD = 1,  /* isGoodSolution after stage 1 (:341)       */
    STVO_PATH_ROBUST_FALLBACK = 2, /* gaussNewtonOptimizationRobust ran (:359) */
    STVO_PATH_REFINED = 4       /* stage 2 ran (:345-350)                    */
};

/* Output of optimizePose.  `T` is what the reference stores in curr_frame->DT, i.e.
 * expmap(logmap(inverse(DT_opt))) (:374); identity when rejected (:385).            */
typedef struct stvo_pose_result {
    double T[16];
    double cov[36];
    double cov_eig[6];
    double err;        /* curr_frame->err_norm; -1 when rejected (:387) */
    double T_opt[16];  /* raw optimiser variable DT before the commit step */
    double err_opt;    /* raw err of the last optimiser call */
    int32_t status;    /* STVO_POSE_* */
    int32_t path;      /* STVO_PATH_* flags */
    int32_t iters[2];  /* optimizeFunctions evaluations in stage 1 / stage 2 (or fallback) */
    int32_t n_matched_pt, n_matched_ls;
    int32_t n_inliers_pt, n_inliers_ls;
} stvo_pose_result;

/* Correspondence records handed to optimizePose: the live fields of matched_pt / matched_ls
 * (include/stereoFeatures.h:30-121).  Host pointers for the host-buffer
 * entry points.  `inlier_*` are in/out (1 = inlier). */
typedef struct stvo_matched {
    int32_t np;
    const double* P;       /* [np][3]  PointFeature::P                                   */
    const double* pl_obs;  /* [np][2]  PointFeature::pl_obs                              */
    const double* sigma2p; /* [np]     PointFeature::sigma2                              */
    int32_t* inlier_p;     /* [np]                                                       */
    int32_t nl;
    const double* sP;      /* [nl][3]  LineFeature::sP                                   */
    const double* eP;      /* [nl][3]  LineFeature::eP                                   */
    const double* le_obs;  /* [nl][3]  LineFeature::le_obs                               */
    const double* spl;     /* [nl][2]  LineFeature::spl (prev-frame end point, overlap)  */
    const double* epl;     /* [nl][2]  LineFeature::epl                                  */
    const double* sigma2l; /* [nl]     LineFeature::sigma2 AFTER safeCopy's re-scaling   */
    int32_t* inlier_l;     /* [nl]                                                       */
} stvo_matched;

/* Extracted features of ONE frame for B independent sequences: what detectStereoPoints / detectStereoLineSegments
 * leave behind before the stereo association (src/stereoFrame.cpp:88-100,191-203).  Host pointers; per-sequence
 * arrays are strided by stride_kp / stride_kl rows.  Key-points are cv::KeyPoint::pt (float x, y) + octave;
 * key-lines are line_descriptor::KeyLine start/end points (float) + octave; descriptors are 32-byte rows. */
typedef struct stvo_frame_features {
    int32_t stride_kp, stride_kl;
    const int32_t* n_kp_l;   /* [B] */
    const int32_t* n_kp_r;   /* [B] */
    const float* kp_l;       /* [B][stride_kp][2] */
    const int32_t* oct_l;    /* [B][stride_kp]    */
    const uint8_t* desc_l;   /* [B][stride_kp][32] */
    const float* kp_r;       /* [B][stride_kp][2] */
    const uint8_t* desc_r;   /* [B][stride_kp][32] */
    const int32_t* n_kl_l;   /* [B] (may be NULL: no lines) */
    const int32_t* n_kl_r;   /* [B] */
    const float* kl_l;       /* [B][stride_kl][4]  sx, sy, ex, ey */
    const int32_t* oct_ll;   /* [B][stride_kl]    */
    const uint8_t* ldesc_l;  /* [B][stride_kl][32] */
    const float* kl_r;       /* [B][stride_kl][4] */
    const uint8_t* ldesc_r;  /* [B][stride_kl][32] */
} stvo_frame_features;

/* Parameters of the ORB point front-end = the cv::ORB::create arguments the reference passes (src/stereoFrame.cpp:112-114)
 * that vary between its configurations; fixed here: WTA_K 2, FAST_SCORE ranking (orb_score 1), patch size 31. */
#define STVO_ORB_MAX_LEVELS 8
typedef struct stvo_orb_params {
    int32_t nfeatures;       /* Config::orbNFeatures()  (2000 in config_kitti.yaml)                     */
    int32_t fast_threshold;  /* Config::orbFastTh() or the handler's adaptive orb_fast_th (1 .. 254)     */
    int32_t edge_threshold;  /* Config::orbEdgeTh()  (19; must be >= 19: patch radius 15, pattern reach) */
    int32_t nlevels;         /* Config::orbNLevels()  (1 in config_kitti.yaml, 4 in config_euroc.yaml:61 and src/config.cpp:97);
                                0 is read as 1; at most STVO_ORB_MAX_LEVELS */
    double scale_factor;     /* Config::orbScaleFactor()  (1.2; > 1; ignored for one level) */
} stvo_orb_params;

/* A key-line as the LBD descriptor consumes it: the line_descriptor::KeyLine fields BinaryDescriptor::computeImpl copies into its
 * OctaveSingleLine (3rdparty/line_descriptor/src/binary_descriptor_custom.cpp:592-609), octave 0. */
typedef struct stvo_keyline {
    float sx, sy, ex, ey;  /* sPointInOctaveX / Y, ePointInOctaveX / Y */
    float angle;           /* KeyLine::angle = atan2(ey - sy, ex - sx), radians */
    int32_t num_pixels;    /* KeyLine::numOfPixels (cv::LineIterator count): the length of the line support region */
} stvo_keyline;

/* Error codes of the C-ABI (0 ok, <0 error; never throws). */
enum {
    STVO_OK = 0,
    STVO_ERR_INVALID_ARG = -1,   /* contract violations the reference throws on (matching.cpp:50,113,184) */
    STVO_ERR_HIP = -2,           /* a HIP runtime call failed */
    STVO_ERR_NO_DEVICE = -3,     /* no gfx950 device visible: the product never falls back to CPU */
    STVO_ERR_CAPACITY = -4,      /* problem larger than the context was created for */
    STVO_ERR_UNSUPPORTED = -5
};

#ifdef __cplusplus
}
#endif
#endif /* STVO_TYPES_H */
