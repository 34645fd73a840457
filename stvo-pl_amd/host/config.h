// config.h — StVO::Config with the reference's accessor names (include/config.h:36-105) and defaults
// (src/config.cpp:36-113).  Meyers singleton with mutable references, as upstream.  The YAML/Boost
// loader is replaced by a `key : value` line reader that accepts the reference's config files
// (config/config/*.yaml are flat `key : value  # comment` lists).
#pragma once
#include <string>

namespace StVO {

class Config {
public:
    static Config& getInstance();
    static void loadFromFile(const std::string& path);  // throws std::runtime_error if unreadable
    static void setKittiPreset();                        // values of config/config/config_kitti.yaml
    static void setEurocPreset();                        // values of config/config/config_euroc.yaml
    static void setDefaults();

#define STVO_CFG(type, name, field) static type& name() { return getInstance().field; }
    STVO_CFG(bool, hasPoints, has_points) STVO_CFG(bool, hasLines, has_lines) STVO_CFG(bool, useFLDLines, use_fld_lines)
    STVO_CFG(bool, lrInParallel, lr_in_parallel) STVO_CFG(bool, plInParallel, pl_in_parallel)
    STVO_CFG(bool, bestLRMatches, best_lr_matches) STVO_CFG(bool, adaptativeFAST, adaptative_fast)
    STVO_CFG(bool, useMotionModel, use_motion_model)
    STVO_CFG(double, maxDistEpip, max_dist_epip) STVO_CFG(double, minDisp, min_disp) STVO_CFG(double, minRatio12P, min_ratio_12_p)
    STVO_CFG(double, lineSimTh, line_sim_th) STVO_CFG(double, stereoOverlapTh, stereo_overlap_th)
    STVO_CFG(double, f2fOverlapTh, f2f_overlap_th) STVO_CFG(double, minLineLength, min_line_length)
    STVO_CFG(double, lineHorizTh, line_horiz_th) STVO_CFG(double, minRatio12L, min_ratio_12_l)
    STVO_CFG(double, lsMinDispRatio, ls_min_disp_ratio)
    STVO_CFG(int, fastMinTh, fast_min_th) STVO_CFG(int, fastMaxTh, fast_max_th) STVO_CFG(int, fastIncTh, fast_inc_th)
    STVO_CFG(int, fastFeatTh, fast_feat_th) STVO_CFG(double, fastErrTh, fast_err_th)
    STVO_CFG(double, homogTh, homog_th) STVO_CFG(int, minFeatures, min_features) STVO_CFG(int, maxIters, max_iters)
    STVO_CFG(int, maxItersRef, max_iters_ref) STVO_CFG(double, minError, min_error)
    STVO_CFG(double, minErrorChange, min_error_change) STVO_CFG(double, inlierK, inlier_k)
    STVO_CFG(int, matchingStrategy, matching_strategy) STVO_CFG(int, matchingSWs, matching_s_ws)
    STVO_CFG(int, matchingF2FWs, matching_f2f_ws)
    STVO_CFG(int, orbNFeatures, orb_nfeatures) STVO_CFG(double, orbScaleFactor, orb_scale_factor)
    STVO_CFG(int, orbNLevels, orb_nlevels) STVO_CFG(int, orbFastTh, orb_fast_th) STVO_CFG(int, orbEdgeTh, orb_edge_th)
    STVO_CFG(int, lsdNFeatures, lsd_nfeatures) STVO_CFG(double, lsdScale, lsd_scale)
    STVO_CFG(int, lsdRefine, lsd_refine) STVO_CFG(double, lsdSigmaScale, lsd_sigma_scale) STVO_CFG(double, lsdQuant, lsd_quant)
    STVO_CFG(double, lsdAngTh, lsd_ang_th) STVO_CFG(double, lsdLogEps, lsd_log_eps) STVO_CFG(double, lsdDensityTh, lsd_density_th)
    STVO_CFG(int, lsdNBins, lsd_n_bins)
    STVO_CFG(double, minEntropyRatio, min_entropy_ratio) STVO_CFG(double, maxKFTDist, max_kf_t_dist)
    STVO_CFG(double, maxKFRDist, max_kf_r_dist)
#undef STVO_CFG

    bool has_points, has_lines, use_fld_lines, lr_in_parallel, pl_in_parallel, best_lr_matches, adaptative_fast,
        use_motion_model;
    double max_dist_epip, min_disp, min_ratio_12_p, line_sim_th, stereo_overlap_th, f2f_overlap_th, min_line_length,
        line_horiz_th, min_ratio_12_l, ls_min_disp_ratio;
    int fast_min_th, fast_max_th, fast_inc_th, fast_feat_th;
    double fast_err_th;
    double homog_th;
    int min_features, max_iters, max_iters_ref;
    double min_error, min_error_change, inlier_k;
    int matching_strategy, matching_s_ws, matching_f2f_ws;
    int orb_nfeatures;
    double orb_scale_factor;
    int orb_nlevels, orb_fast_th, orb_edge_th;
    int lsd_nfeatures;
    double lsd_scale;
    int lsd_refine, lsd_n_bins;  // src/config.cpp:105-112
    double lsd_sigma_scale, lsd_quant, lsd_ang_th, lsd_log_eps, lsd_density_th;
    double min_entropy_ratio, max_kf_t_dist, max_kf_r_dist;

private:
    Config();
};

}  // namespace StVO
