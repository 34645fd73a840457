"""ctypes mirrors of include/stvo_types.h (POD records of the C-ABI)."""
import ctypes as C

import numpy as np


class Cam(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("b", C.c_double)]

    @classmethod
    def from_dict(cls, d):
        return cls(d["fx"], d["fy"], d["cx"], d["cy"], d["b"])


class GridWindow(C.Structure):
    _fields_ = [("w_lo", C.c_int32), ("w_hi", C.c_int32), ("h_lo", C.c_int32), ("h_hi", C.c_int32)]


class OptParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("has_points", C.c_int32), ("has_lines", C.c_int32), ("min_features", C.c_int32),
                ("max_iters", C.c_int32), ("max_iters_ref", C.c_int32), ("reserved0", C.c_int32),
                ("reserved1", C.c_int32), ("homog_th", C.c_double), ("min_error", C.c_double),
                ("min_error_change", C.c_double), ("inlier_k", C.c_double)]


class MatchParams(C.Structure):
    _fields_ = [("best_lr_matches", C.c_int32), ("matching_s_ws", C.c_int32), ("min_ratio_12_p", C.c_float),
                ("min_ratio_12_l", C.c_float), ("max_dist_epip", C.c_double), ("min_disp", C.c_double),
                ("line_sim_th", C.c_double), ("stereo_overlap_th", C.c_double), ("line_horiz_th", C.c_double),
                ("ls_min_disp_ratio", C.c_double), ("orb_scale_factor", C.c_double), ("lsd_scale", C.c_double),
                ("min_ratio_12_p_d", C.c_double)]


class PoseResult(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("cov", C.c_double * 36), ("cov_eig", C.c_double * 6), ("err", C.c_double),
                ("T_opt", C.c_double * 16), ("err_opt", C.c_double), ("status", C.c_int32), ("path", C.c_int32),
                ("iters", C.c_int32 * 2), ("n_matched_pt", C.c_int32), ("n_matched_ls", C.c_int32),
                ("n_inliers_pt", C.c_int32), ("n_inliers_ls", C.c_int32)]

    def as_dict(self):
        return dict(T=np.array(self.T).reshape(4, 4), cov=np.array(self.cov).reshape(6, 6),
                    cov_eig=np.array(self.cov_eig), err=self.err, T_opt=np.array(self.T_opt).reshape(4, 4),
                    err_opt=self.err_opt, status=self.status, path=self.path, iters=(self.iters[0], self.iters[1]),
                    n_matched_pt=self.n_matched_pt, n_matched_ls=self.n_matched_ls, n_inliers_pt=self.n_inliers_pt,
                    n_inliers_ls=self.n_inliers_ls)


POSE_RESULT_DTYPE = np.dtype([("T", "<f8", (16,)), ("cov", "<f8", (36,)), ("cov_eig", "<f8", (6,)), ("err", "<f8"),
                              ("T_opt", "<f8", (16,)), ("err_opt", "<f8"), ("status", "<i4"), ("path", "<i4"),
                              ("iters", "<i4", (2,)), ("n_matched_pt", "<i4"), ("n_matched_ls", "<i4"),
                              ("n_inliers_pt", "<i4"), ("n_inliers_ls", "<i4")])
assert POSE_RESULT_DTYPE.itemsize == C.sizeof(PoseResult), (POSE_RESULT_DTYPE.itemsize, C.sizeof(PoseResult))

STATUS_OK, STATUS_FEW_BEFORE, STATUS_FEW_AFTER, STATUS_REJECTED = 0, 1, 2, 3
PATH_STAGE1_GOOD, PATH_ROBUST_FALLBACK, PATH_REFINED = 1, 2, 4


# Parameter presets = the values of the reference's YAML files.
def opt_params(preset="kitti", **kw):
    base = dict(mode=0, has_points=1, has_lines=1, min_features=10, max_iters=5, max_iters_ref=10, homog_th=1e-7,
                min_error=1e-7, min_error_change=1e-7, inlier_k=4.0)  # src/config.cpp:80-86
    if preset == "kitti":
        base["inlier_k"] = 1.2  # config/config/config_kitti.yaml:45
    elif preset == "euroc":
        base["inlier_k"] = 4.0  # config/config/config_euroc.yaml:50
    elif preset != "default":
        raise ValueError(preset)
    base.update(kw)
    return OptParams(base["mode"], base["has_points"], base["has_lines"], base["min_features"], base["max_iters"],
                     base["max_iters_ref"], 0, 0, base["homog_th"], base["min_error"], base["min_error_change"],
                     base["inlier_k"])


def match_params(preset="kitti", **kw):
    base = dict(best_lr_matches=1, matching_s_ws=10, min_ratio_12_p=0.9, min_ratio_12_l=0.9, max_dist_epip=1.0,
                min_disp=1.0, line_sim_th=0.75, stereo_overlap_th=0.75, line_horiz_th=0.1, ls_min_disp_ratio=0.7,
                orb_scale_factor=1.2, lsd_scale=1.2)  # src/config.cpp:49-69,91,96,106
    if preset == "kitti":
        base.update(min_ratio_12_p=0.75, min_ratio_12_l=0.75, max_dist_epip=0.0)  # config_kitti.yaml:17,19,27
    elif preset == "euroc":
        pass  # config_euroc.yaml:22,24,32 equal the defaults
    elif preset != "default":
        raise ValueError(preset)
    base.update(kw)
    return MatchParams(base["best_lr_matches"], base["matching_s_ws"], base["min_ratio_12_p"], base["min_ratio_12_l"],
                       base["max_dist_epip"], base["min_disp"], base["line_sim_th"], base["stereo_overlap_th"],
                       base["line_horiz_th"], base["ls_min_disp_ratio"], base["orb_scale_factor"], base["lsd_scale"],
                       float(base["min_ratio_12_p"]))  # matchGrid compares with the double (src/matching.cpp:160,241)
