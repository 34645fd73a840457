#include "config.h"

#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

namespace StVO {

Config::Config() { /* filled by setDefaults() */ }

Config& Config::getInstance() {
    static Config instance;
    static bool init = false;
    if (!init) {
        init = true;
        setDefaults();
    }
    return instance;
}

// src/config.cpp:36-113
void Config::setDefaults() {
    Config& c = getInstance();
    c.min_entropy_ratio = 0.85; c.max_kf_t_dist = 5.0; c.max_kf_r_dist = 15.0;
    c.has_points = true; c.has_lines = true; c.use_fld_lines = false; c.lr_in_parallel = true; c.pl_in_parallel = true;
    c.best_lr_matches = true; c.adaptative_fast = true; c.use_motion_model = false;
    c.max_dist_epip = 1.0; c.min_disp = 1.0; c.min_ratio_12_p = 0.9;
    c.line_sim_th = 0.75; c.stereo_overlap_th = 0.75; c.f2f_overlap_th = 0.75; c.min_line_length = 0.025;
    c.line_horiz_th = 0.1; c.min_ratio_12_l = 0.9; c.ls_min_disp_ratio = 0.7;
    c.fast_min_th = 5; c.fast_max_th = 50; c.fast_inc_th = 5; c.fast_feat_th = 50; c.fast_err_th = 0.5;
    c.homog_th = 1e-7; c.min_features = 10; c.max_iters = 5; c.max_iters_ref = 10; c.min_error = 1e-7;
    c.min_error_change = 1e-7; c.inlier_k = 4.0;
    c.matching_strategy = 0; c.matching_s_ws = 10; c.matching_f2f_ws = 3;
    c.orb_nfeatures = 1200; c.orb_scale_factor = 1.2; c.orb_nlevels = 4; c.orb_fast_th = 20; c.orb_edge_th = 19;  // src/config.cpp:95-102
    c.lsd_nfeatures = 300; c.lsd_scale = 1.2;
    c.lsd_refine = 0; c.lsd_sigma_scale = 0.6; c.lsd_quant = 2.0; c.lsd_ang_th = 22.5; c.lsd_log_eps = 1.0; c.lsd_density_th = 0.6;
    c.lsd_n_bins = 1024;  // src/config.cpp:105-112
}

// config/config/config_kitti.yaml
void Config::setKittiPreset() {
    setDefaults();
    Config& c = getInstance();
    c.max_dist_epip = 0.0; c.min_ratio_12_p = 0.75; c.min_ratio_12_l = 0.75; c.inlier_k = 1.2;
    c.fast_min_th = 7; c.fast_max_th = 30; c.orb_nfeatures = 2000; c.orb_nlevels = 1; c.lsd_nfeatures = 100;
}

// config/config/config_euroc.yaml (the values the path reads equal the defaults; 800 ORB / 300 LSD)
void Config::setEurocPreset() {
    setDefaults();
    Config& c = getInstance();
    c.orb_nfeatures = 800; c.orb_nlevels = 4; c.lsd_nfeatures = 300;
}

namespace {
bool parse_bool(const std::string& v) { return v == "true" || v == "True" || v == "1" || v == "yes"; }
}

// Accepts `key : value   # comment` lines; unknown keys are ignored and missing keys keep their
// current value (the reference's loadSafe fallback, src/config.cpp:123-130).
void Config::loadFromFile(const std::string& path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("[Config] cannot open " + path);
    std::map<std::string, std::string> kv;
    std::string line;
    while (std::getline(in, line)) {
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        const size_t colon = line.find(':');
        if (colon == std::string::npos) continue;
        auto trim = [](std::string s) {
            const char* ws = " \t\r\n";
            const size_t b = s.find_first_not_of(ws);
            if (b == std::string::npos) return std::string();
            return s.substr(b, s.find_last_not_of(ws) - b + 1);
        };
        const std::string k = trim(line.substr(0, colon)), v = trim(line.substr(colon + 1));
        if (!k.empty() && !v.empty()) kv[k] = v;
    }
    Config& c = getInstance();
#define B(key, field) if (kv.count(key)) c.field = parse_bool(kv[key]);
#define D(key, field) if (kv.count(key)) c.field = std::atof(kv[key].c_str());
#define I(key, field) if (kv.count(key)) c.field = std::atoi(kv[key].c_str());
    B("has_points", has_points) B("has_lines", has_lines) B("use_fld_lines", use_fld_lines) B("lr_in_parallel", lr_in_parallel)
    B("pl_in_parallel", pl_in_parallel) B("best_lr_matches", best_lr_matches) B("adaptative_fast", adaptative_fast)
    B("use_motion_model", use_motion_model)
    D("max_dist_epip", max_dist_epip) D("min_disp", min_disp) D("min_ratio_12_p", min_ratio_12_p) D("line_sim_th", line_sim_th)
    D("stereo_overlap_th", stereo_overlap_th) D("f2f_overlap_th", f2f_overlap_th) D("min_line_length", min_line_length)
    D("line_horiz_th", line_horiz_th) D("min_ratio_12_l", min_ratio_12_l) D("ls_min_disp_ratio", ls_min_disp_ratio)
    I("fast_min_th", fast_min_th) I("fast_max_th", fast_max_th) I("fast_inc_th", fast_inc_th) I("fast_feat_th", fast_feat_th)
    D("fast_err_th", fast_err_th) D("homog_th", homog_th) I("min_features", min_features) I("max_iters", max_iters)
    I("max_iters_ref", max_iters_ref) D("min_error", min_error) D("min_error_change", min_error_change) D("inlier_k", inlier_k)
    I("matching_strategy", matching_strategy)  // NB: config_kitti.yaml's `matching_stereo` is not read upstream either
    I("matching_s_ws", matching_s_ws) I("matching_f2f_ws", matching_f2f_ws)
    I("orb_nfeatures", orb_nfeatures) D("orb_scale_factor", orb_scale_factor) I("orb_nlevels", orb_nlevels)
    I("orb_fast_th", orb_fast_th) I("orb_edge_th", orb_edge_th) I("lsd_nfeatures", lsd_nfeatures) D("lsd_scale", lsd_scale)
    I("lsd_refine", lsd_refine) D("lsd_sigma_scale", lsd_sigma_scale) D("lsd_quant", lsd_quant) D("lsd_ang_th", lsd_ang_th)
    D("lsd_log_eps", lsd_log_eps) D("lsd_density_th", lsd_density_th) I("lsd_n_bins", lsd_n_bins)
    D("min_entropy_ratio", min_entropy_ratio) D("max_kf_t_dist", max_kf_t_dist) D("max_kf_r_dist", max_kf_r_dist)
#undef B
#undef D
#undef I
}

}  // namespace StVO
