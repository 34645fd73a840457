"""ctypes binding of the product library stvo-pl_amd/libstvo_hip.so (include/stvo_hip.h).

Plumbing for tests and bench only.  There is NO fallback: if the library is missing or no gfx950
device is visible, loading / context creation raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .ctypes_types import Cam, GridWindow, MatchParams, OptParams, PoseResult, POSE_RESULT_DTYPE

PKG_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # stvo-pl_amd/
LIB_PATH = os.environ.get("STVO_LIB") or os.path.join(PKG_DIR, "libstvo_hip.so")   # STVO_LIB: A/B runs of two builds (developer)

EXPORTS = ["stvo_backend_name", "stvo_abi_version", "stvo_error_string", "stvo_ctx_last_error", "stvo_ctx_create",
           "stvo_ctx_destroy", "stvo_ctx_set_stream", "stvo_ctx_synchronize", "stvo_ctx_set_overlap", "stvo_match_nnr_mutual",
           "stvo_match_grid_points", "stvo_match_grid_lines", "stvo_normal_eq", "stvo_optimize_pose",
           "stvo_track_batched_dev", "stvo_match_nnr_mutual_batched_dev", "stvo_optimize_pose_batched_dev",
           "stvo_time_stage_dev", "stvo_valu_peak_probe", "stvo_last_reverse_counts", "stvo_last_reverse_plan", "stvo_ctx_set_kernel_timing", "stvo_ctx_get_kernel_timing", "stvo_seq_create", "stvo_seq_destroy", "stvo_seq_enable_fetch", "stvo_seq_fetch_matches", "stvo_seq_fetch_inliers", "stvo_seq_strides",
           "stvo_seq_push", "stvo_seq_upload", "stvo_seq_step_dev", "stvo_seq_read", "stvo_seq_create_multi", "stvo_seq_set_slots",
           "stvo_seq_set_stage_timing", "stvo_seq_get_stage_timing", "stvo_seq_set_motion_model", "stvo_seq_debug_grid", "stvo_orb_create", "stvo_orb_destroy",
           "stvo_orb_set_pattern", "stvo_orb_get_pattern", "stvo_orb_detect", "stvo_orb_detect_dev", "stvo_orb_detect_levels",
           "stvo_orb_detect_levels_dev", "stvo_orb_set_fast_threshold", "stvo_seq_upload_dev", "stvo_lbd_create", "stvo_lbd_destroy",
           "stvo_lbd_compute", "stvo_lbd_compute_dev", "stvo_debug_reparse_env", "stvo_lsd_create", "stvo_lsd_destroy", "stvo_lsd_detect",
           "stvo_lsd_detect_dev", "stvo_lsd_segments", "stvo_lsd_counts", "stvo_keylines_xy_dev"]

SEQ_NSTAGE = 5  # include/stvo_hip.h: STVO_SEQ_NSTAGE
SEQ_STAGE_NAMES = ("stereo_points_stage", "grid_scan", "hamming_knn2", "reverse_check", "pose")

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


class Matched(C.Structure):
    _fields_ = [("np", C.c_int32), ("P", C.c_void_p), ("pl_obs", C.c_void_p), ("sigma2p", C.c_void_p),
                ("inlier_p", C.c_void_p), ("nl", C.c_int32), ("sP", C.c_void_p), ("eP", C.c_void_p),
                ("le_obs", C.c_void_p), ("spl", C.c_void_p), ("epl", C.c_void_p), ("sigma2l", C.c_void_p),
                ("inlier_l", C.c_void_p)]


class TrackBatchDev(C.Structure):
    _fields_ = [("B", C.c_int32), ("max_pts", C.c_int32), ("max_lines", C.c_int32), ("reserved", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("n_prev_pts", "prev_pdesc", "prev_P", "prev_sigma2p", "n_curr_pts",
                                          "curr_pdesc", "curr_pl", "n_prev_lines", "prev_ldesc", "prev_sP", "prev_eP",
                                          "prev_spl", "prev_epl", "prev_sigma2l", "n_curr_lines", "curr_ldesc",
                                          "curr_le", "init_T", "m12_pts", "m12_lines", "inlier_pts", "inlier_lines",
                                          "results")]


class FrameFeatures(C.Structure):
    _fields_ = [("stride_kp", C.c_int32), ("stride_kl", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("n_kp_l", "n_kp_r", "kp_l", "oct_l", "desc_l", "kp_r", "desc_r", "n_kl_l", "n_kl_r",
                                          "kl_l", "oct_ll", "ldesc_l", "kl_r", "ldesc_r")]


class StvoError(RuntimeError):
    pass


def build(force=False):
    """Compile libstvo_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", PKG_DIR, "clean"])
    subprocess.check_call(["make", "-s", "-C", PKG_DIR, "-j4"])
    return LIB_PATH


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StvoError(f"{LIB_PATH} is missing: run `make -C stvo-pl_amd` (python __graft_entry__.py). "
                        "The product has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; if this library pulled in
    # /opt/rocm's copy first, torch would later find "No HIP GPUs".  Import torch first when it exists so
    # both resolve to the same already-loaded runtime (torch is only plumbing: HBM buffers + streams).
    try:
        import torch  # noqa: F401
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.stvo_backend_name.restype = C.c_char_p
    L.stvo_error_string.restype = C.c_char_p
    L.stvo_error_string.argtypes = [C.c_int]
    L.stvo_ctx_last_error.restype = C.c_char_p
    L.stvo_ctx_last_error.argtypes = [C.c_void_p]
    L.stvo_ctx_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.stvo_ctx_destroy.argtypes = [C.c_void_p]
    L.stvo_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.stvo_ctx_synchronize.argtypes = [C.c_void_p]
    L.stvo_ctx_set_overlap.argtypes = [C.c_void_p, C.c_int]
    L.stvo_match_nnr_mutual.argtypes = [C.c_void_p, u8p, C.c_int, u8p, C.c_int, C.c_float, C.c_int, i32p,
                                        C.POINTER(C.c_int32)]
    L.stvo_match_grid_points.argtypes = [C.c_void_p, i32p, u8p, C.c_int, i32p, i32p, u8p, C.c_int,
                                         C.POINTER(GridWindow), C.c_double, C.c_int, i32p, C.POINTER(C.c_int32)]
    L.stvo_match_grid_lines.argtypes = [C.c_void_p, i32p, u8p, C.c_int, i32p, i32p, u8p, C.c_int, f64p,
                                        C.POINTER(GridWindow), C.c_double, C.c_double, C.c_int, i32p,
                                        C.POINTER(C.c_int32)]
    L.stvo_normal_eq.argtypes = [C.c_void_p, f64p, C.POINTER(Cam), C.POINTER(OptParams), C.POINTER(Matched), C.c_int,
                                 f64p, f64p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.stvo_optimize_pose.argtypes = [C.c_void_p, f64p, C.POINTER(Cam), C.POINTER(OptParams), C.POINTER(Matched),
                                     C.POINTER(PoseResult)]
    L.stvo_track_batched_dev.argtypes = [C.c_void_p, C.POINTER(TrackBatchDev), C.POINTER(Cam), C.POINTER(OptParams),
                                         C.c_float, C.c_float, C.c_int]
    L.stvo_match_nnr_mutual_batched_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    L.stvo_optimize_pose_batched_dev.argtypes = [C.c_void_p, C.POINTER(TrackBatchDev), C.POINTER(Cam),
                                                 C.POINTER(OptParams)]
    L.stvo_time_stage_dev.argtypes = [C.c_void_p, C.POINTER(TrackBatchDev), C.POINTER(Cam), C.POINTER(OptParams),
                                      C.c_float, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.stvo_valu_peak_probe.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.stvo_seq_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Cam),
                                  C.POINTER(MatchParams), C.POINTER(OptParams), C.POINTER(C.c_void_p)]
    L.stvo_seq_create_multi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, i32p, i32p, C.c_void_p, C.POINTER(MatchParams),
                                        C.POINTER(OptParams), C.POINTER(C.c_void_p)]
    L.stvo_seq_set_slots.argtypes = [C.c_void_p, C.c_int]
    L.stvo_seq_set_stage_timing.argtypes = [C.c_void_p, C.c_int]
    L.stvo_seq_set_motion_model.argtypes = [C.c_void_p, C.c_int]
    L.stvo_seq_get_stage_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    L.stvo_seq_debug_grid.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, i32p, C.c_int32, i32p, i32p, i32p, C.c_int32,
                                      C.POINTER(C.c_int32)]
    L.stvo_orb_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(OrbParams), C.POINTER(C.c_void_p)]
    L.stvo_orb_destroy.argtypes = [C.c_void_p]
    i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    L.stvo_orb_set_pattern.argtypes = [C.c_void_p, i8p]
    L.stvo_orb_get_pattern.argtypes = [C.c_void_p, i8p]
    L.stvo_orb_detect.argtypes = [C.c_void_p, u8p, f32p, f32p, f32p, u8p, i32p]
    L.stvo_orb_detect_dev.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    L.stvo_orb_set_fast_threshold.argtypes = [C.c_void_p, C.c_int]
    L.stvo_orb_detect_levels.argtypes = [C.c_void_p, u8p, f32p, f32p, f32p, i32p, u8p, i32p, i32p]
    L.stvo_orb_detect_levels_dev.argtypes = [C.c_void_p] + [C.c_void_p] * 8
    L.stvo_lbd_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.stvo_lbd_destroy.argtypes = [C.c_void_p]
    L.stvo_lbd_compute.argtypes = [C.c_void_p, u8p, C.c_void_p, i32p, u8p, C.c_void_p]
    L.stvo_lbd_compute_dev.argtypes = [C.c_void_p] + [C.c_void_p] * 5
    L.stvo_lsd_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(LsdParams), C.POINTER(C.c_void_p)]
    L.stvo_lsd_destroy.argtypes = [C.c_void_p]
    L.stvo_lsd_detect.argtypes = [C.c_void_p, u8p, C.c_void_p, C.c_void_p, i32p]
    L.stvo_lsd_detect_dev.argtypes = [C.c_void_p] + [C.c_void_p] * 4
    L.stvo_lsd_segments.argtypes = [C.c_void_p, u8p, f32p, C.c_int, i32p]
    L.stvo_lsd_counts.argtypes = [C.c_void_p, i32p, i32p]
    L.stvo_keylines_xy_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.stvo_seq_destroy.argtypes = [C.c_void_p]
    L.stvo_seq_push.argtypes = [C.c_void_p, C.POINTER(FrameFeatures), C.c_void_p, i32p]
    L.stvo_seq_upload.argtypes = [C.c_void_p, C.c_int, C.POINTER(FrameFeatures)]
    L.stvo_seq_upload_dev.argtypes = [C.c_void_p, C.c_int, C.POINTER(FrameFeatures)]
    L.stvo_seq_step_dev.argtypes = [C.c_void_p, C.c_int]
    L.stvo_seq_read.argtypes = [C.c_void_p, C.c_void_p, i32p]
    pp32 = C.POINTER(C.POINTER(C.c_int32))
    L.stvo_seq_enable_fetch.argtypes = [C.c_void_p, C.c_int]
    L.stvo_seq_fetch_matches.argtypes = [C.c_void_p, pp32, pp32, pp32, pp32]
    L.stvo_seq_fetch_inliers.argtypes = [C.c_void_p, pp32, pp32]
    L.stvo_seq_strides.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.stvo_last_reverse_counts.argtypes = [C.c_void_p, C.c_int, i32p]
    L.stvo_last_reverse_plan.argtypes = [C.c_void_p, C.c_int, i32p]
    L.stvo_ctx_set_kernel_timing.argtypes = [C.c_void_p, C.c_int]
    L.stvo_ctx_get_kernel_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One stvo_ctx.  Raises StvoError on any non-zero status."""

    def __init__(self, device_id=0, max_rows=4096, max_batch=64):
        self.lib = load()
        self.h = C.c_void_p()
        rc = self.lib.stvo_ctx_create(device_id, max_rows, max_batch, C.byref(self.h))
        if rc != 0:
            self.h = None
            raise StvoError(f"stvo_ctx_create failed: {self.lib.stvo_error_string(rc).decode()} ({rc})")

    def close(self):
        if self.h:
            self.lib.stvo_ctx_destroy(self.h)
            self.h = None

    def _chk(self, rc):
        if rc != 0:
            raise StvoError(f"{self.lib.stvo_error_string(rc).decode()} ({rc}): "
                            f"{self.lib.stvo_ctx_last_error(self.h).decode()}")

    def set_stream(self, stream_handle):
        self._chk(self.lib.stvo_ctx_set_stream(self.h, C.c_void_p(stream_handle)))

    def synchronize(self):
        self._chk(self.lib.stvo_ctx_synchronize(self.h))

    def set_overlap(self, enable):
        self._chk(self.lib.stvo_ctx_set_overlap(self.h, 1 if enable else 0))

    # ---- host-buffer seams ----
    def match(self, d1, d2, nnr, mutual=1):
        d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
        m12 = np.empty(max(len(d1), 1), np.int32)
        n = C.c_int32()
        self._chk(self.lib.stvo_match_nnr_mutual(self.h, d1 if len(d1) else np.zeros((1, 32), np.uint8), len(d1),
                                                 d2 if len(d2) else np.zeros((1, 32), np.uint8), len(d2), nnr, mutual,
                                                 m12, C.byref(n)))
        return m12[:len(d1)], n.value

    def match_grid_points(self, cell_xy1, d1, start, items, d2, w, ratio, mutual=1):
        n1 = len(d1)
        m12 = np.empty(max(n1, 1), np.int32)
        n = C.c_int32()
        gw = GridWindow(*w)
        items_ = np.ascontiguousarray(items, np.int32) if len(items) else np.zeros(1, np.int32)
        d1_ = d1 if n1 else np.zeros((1, 32), np.uint8)
        d2_ = d2 if len(d2) else np.zeros((1, 32), np.uint8)
        xy = np.ascontiguousarray(cell_xy1, np.int32).reshape(-1) if n1 else np.zeros(2, np.int32)
        self._chk(self.lib.stvo_match_grid_points(self.h, xy, d1_, n1, np.ascontiguousarray(start, np.int32), items_,
                                                  d2_, len(d2), C.byref(gw), ratio, mutual, m12, C.byref(n)))
        return m12[:n1], n.value

    def match_grid_lines(self, cell_xy1, d1, start, items, d2, dir2, w, ratio, line_sim_th, mutual=1):
        n1 = len(d1)
        m12 = np.empty(max(n1, 1), np.int32)
        n = C.c_int32()
        gw = GridWindow(*w)
        items_ = np.ascontiguousarray(items, np.int32) if len(items) else np.zeros(1, np.int32)
        d1_ = d1 if n1 else np.zeros((1, 32), np.uint8)
        d2_ = d2 if len(d2) else np.zeros((1, 32), np.uint8)
        xy = np.ascontiguousarray(cell_xy1, np.int32).reshape(-1) if n1 else np.zeros(4, np.int32)
        dr = np.ascontiguousarray(dir2, np.float64).reshape(-1) if len(d2) else np.zeros(2)
        self._chk(self.lib.stvo_match_grid_lines(self.h, xy, d1_, n1, np.ascontiguousarray(start, np.int32), items_,
                                                 d2_, len(d2), dr, C.byref(gw), ratio, line_sim_th, mutual, m12,
                                                 C.byref(n)))
        return m12[:n1], n.value

    @staticmethod
    def _matched(rec):
        keep = {k: np.ascontiguousarray(rec[k], np.float64)
                for k in ("P", "pl_obs", "sigma2p", "sP", "eP", "le_obs", "spl", "epl", "sigma2l")}
        keep["inlier_p"] = np.ascontiguousarray(rec["inlier_p"], np.int32).copy()
        keep["inlier_l"] = np.ascontiguousarray(rec["inlier_l"], np.int32).copy()
        m = Matched(len(keep["sigma2p"]), _ptr(keep["P"]), _ptr(keep["pl_obs"]), _ptr(keep["sigma2p"]),
                    _ptr(keep["inlier_p"]), len(keep["sigma2l"]), _ptr(keep["sP"]), _ptr(keep["eP"]),
                    _ptr(keep["le_obs"]), _ptr(keep["spl"]), _ptr(keep["epl"]), _ptr(keep["sigma2l"]),
                    _ptr(keep["inlier_l"]))
        return m, keep

    def normal_eq(self, T, cam, params, rec, robust=0):
        m, keep = self._matched(rec)
        H = np.empty(36)
        g = np.empty(6)
        e = C.c_double()
        n = C.c_int32()
        camc = Cam.from_dict(cam)
        self._chk(self.lib.stvo_normal_eq(self.h, np.ascontiguousarray(T, np.float64).reshape(-1), C.byref(camc),
                                          C.byref(params), C.byref(m), robust, H, g, C.byref(e), C.byref(n)))
        return H.reshape(6, 6), g, e.value, n.value

    def optimize_pose(self, init_T, cam, params, rec):
        m, keep = self._matched(rec)
        res = PoseResult()
        camc = Cam.from_dict(cam)
        self._chk(self.lib.stvo_optimize_pose(self.h, np.ascontiguousarray(init_T, np.float64).reshape(-1),
                                              C.byref(camc), C.byref(params), C.byref(m), C.byref(res)))
        d = res.as_dict()
        d["inlier_p"] = keep["inlier_p"]
        d["inlier_l"] = keep["inlier_l"]
        return d

    def valu_peak(self):
        v = C.c_double()
        self._chk(self.lib.stvo_valu_peak_probe(self.h, C.byref(v)))
        return v.value

    def set_kernel_timing(self, enable):
        self._chk(self.lib.stvo_ctx_set_kernel_timing(self.h, 1 if enable else 0))

    def get_kernel_timing(self):
        f = C.c_float(); r = C.c_float(); n = C.c_int32()
        self._chk(self.lib.stvo_ctx_get_kernel_timing(self.h, C.byref(f), C.byref(r), C.byref(n)))
        return f.value, r.value, n.value

    def last_reverse_counts(self, B):
        out = np.empty(B, np.int32)
        self._chk(self.lib.stvo_last_reverse_counts(self.h, B, out))
        return out

    def last_reverse_plan(self, B):
        """[5][B]: claimed, light, heavy columns, |S|, tau of the last mutual match's reverse check (matrix-core path)."""
        out = np.empty((5, B), np.int32)
        self._chk(self.lib.stvo_last_reverse_plan(self.h, B, out))
        return out

    # ---- batched device-resident path ----
    def track_batched(self, batch, cam, params, nnr_p, nnr_l, mutual=1):
        camc = Cam.from_dict(cam)
        self._chk(self.lib.stvo_track_batched_dev(self.h, C.byref(batch.struct), C.byref(camc), C.byref(params), nnr_p,
                                                  nnr_l, mutual))

    def optimize_pose_batched(self, batch, cam, params):
        camc = Cam.from_dict(cam)
        self._chk(self.lib.stvo_optimize_pose_batched_dev(self.h, C.byref(batch.struct), C.byref(camc), C.byref(params)))

    def time_stage(self, batch, cam, params, nnr, stage, iters):
        camc = Cam.from_dict(cam)
        ms = C.c_float()
        self._chk(self.lib.stvo_time_stage_dev(self.h, C.byref(batch.struct), C.byref(camc), C.byref(params), nnr,
                                               stage, iters, C.byref(ms)))
        return ms.value


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("fast_threshold", C.c_int32), ("edge_threshold", C.c_int32), ("nlevels", C.c_int32),
                ("scale_factor", C.c_double)]


class Orb:
    """The ORB point front-end for B images of one size (stvo_orb_*)."""

    def __init__(self, ctx, B, cols, rows, max_keypoints=2048, nfeatures=2000, fast_threshold=20, edge_threshold=19, nlevels=1,
                 scale_factor=1.2):
        self.ctx, self.B, self.cols, self.rows, self.K = ctx, B, cols, rows, max_keypoints
        self.h = C.c_void_p()
        prm = OrbParams(nfeatures, fast_threshold, edge_threshold, nlevels, scale_factor)
        ctx._chk(ctx.lib.stvo_orb_create(ctx.h, B, cols, rows, max_keypoints, C.byref(prm), C.byref(self.h)))

    def close(self):
        if self.h:
            self.ctx.lib.stvo_orb_destroy(self.h)
            self.h = None

    def pattern(self):
        p = np.zeros(1024, np.int8)
        self.ctx._chk(self.ctx.lib.stvo_orb_get_pattern(self.h, p))
        return p.reshape(256, 4)

    def set_pattern(self, pattern):
        self.ctx._chk(self.ctx.lib.stvo_orb_set_pattern(self.h, np.ascontiguousarray(pattern, np.int8).reshape(-1)))

    def set_fast_threshold(self, th):
        self.ctx._chk(self.ctx.lib.stvo_orb_set_fast_threshold(self.h, th))

    def detect(self, images):
        """images: uint8 [B, rows, cols] -> list of B dicts(kp [n,2] float32, response, angle, desc [n,32])."""
        images = np.ascontiguousarray(images, np.uint8).reshape(self.B, self.rows, self.cols)
        kp = np.zeros((self.B, self.K, 2), np.float32); resp = np.zeros((self.B, self.K), np.float32)
        ang = np.zeros((self.B, self.K), np.float32); desc = np.zeros((self.B, self.K, 32), np.uint8); n = np.zeros(self.B, np.int32)
        octv = np.zeros((self.B, self.K), np.int32); ntot = np.zeros(self.B, np.int32)
        self.ctx._chk(self.ctx.lib.stvo_orb_detect_levels(self.h, images.reshape(-1), kp.reshape(-1), resp.reshape(-1), ang.reshape(-1),
                                                          octv.reshape(-1), desc.reshape(-1), n, ntot))
        return [dict(kp=kp[b, :n[b]].copy(), response=resp[b, :n[b]].copy(), angle=ang[b, :n[b]].copy(), desc=desc[b, :n[b]].copy(),
                     octave=octv[b, :n[b]].copy(), n_total=int(ntot[b])) for b in range(self.B)]

    def detect_dev(self, img, kp, resp, ang, desc, n, octave=None, n_total=None):
        """Device buffers (integers = device addresses): images in, key-points / descriptors out, on the context's stream."""
        self.ctx._chk(self.ctx.lib.stvo_orb_detect_levels_dev(self.h, img, kp, resp, ang, octave, desc, n, n_total))


KEYLINE_DTYPE = np.dtype([("sx", "<f4"), ("sy", "<f4"), ("ex", "<f4"), ("ey", "<f4"), ("angle", "<f4"), ("num_pixels", "<i4")])  # stvo_keyline


class LsdParams(C.Structure):  # stvo_lsd_params
    _fields_ = [("refine", C.c_int32), ("n_bins", C.c_int32), ("scale", C.c_double), ("sigma_scale", C.c_double), ("quant", C.c_double),
                ("ang_th", C.c_double), ("log_eps", C.c_double), ("density_th", C.c_double), ("min_length", C.c_double),
                ("nfeatures", C.c_int32), ("reserved", C.c_int32)]


def lsd_params(min_length=0.0, nfeatures=300, refine=0, scale=1.2, sigma_scale=0.6, quant=2.0, ang_th=22.5, n_bins=1024):
    """src/config.cpp:104-112 (every shipped yaml has the same values)."""
    return LsdParams(refine, n_bins, scale, sigma_scale, quant, ang_th, 1.0, 0.6, min_length, nfeatures, 0)


class Lsd:
    """The LSD key-line detector for B images of one size (stvo_lsd_*)."""

    def __init__(self, ctx, B, cols, rows, prm, max_keylines=512):
        self.ctx, self.B, self.cols, self.rows, self.M = ctx, B, cols, rows, max_keylines
        self.h = C.c_void_p()
        ctx._chk(ctx.lib.stvo_lsd_create(ctx.h, B, cols, rows, max_keylines, C.byref(prm), C.byref(self.h)))

    def close(self):
        if self.h:
            self.ctx.lib.stvo_lsd_destroy(self.h)
            self.h = None

    def detect(self, images):
        """images uint8 [B, rows, cols] -> list of B (key-lines as KEYLINE_DTYPE records, responses float32)."""
        images = np.ascontiguousarray(images, np.uint8).reshape(self.B, self.rows, self.cols)
        rec = np.zeros((self.B, self.M), KEYLINE_DTYPE)
        resp = np.zeros((self.B, self.M), np.float32)
        n = np.zeros(self.B, np.int32)
        self.ctx._chk(self.ctx.lib.stvo_lsd_detect(self.h, images.reshape(-1), rec.ctypes.data_as(C.c_void_p), resp.ctypes.data_as(C.c_void_p), n))
        return [(rec[b, :n[b]].copy(), resp[b, :n[b]].copy()) for b in range(self.B)]

    def detect_dev(self, img_ptr, lines_ptr, resp_ptr, n_ptr):
        """Device pointers (uint8 [B, rows, cols]; stvo_keyline [B, M]; float32 [B, M] or None; int32 [B]); asynchronous."""
        self.ctx._chk(self.ctx.lib.stvo_lsd_detect_dev(self.h, img_ptr, lines_ptr, resp_ptr, n_ptr))

    def counts(self):
        """(segments found, segments longer than min_length) per image of the last detection, before the top-N / capacity cut."""
        ns = np.zeros(self.B, np.int32); npass = np.zeros(self.B, np.int32)
        self.ctx._chk(self.ctx.lib.stvo_lsd_counts(self.h, ns, npass))
        return ns, npass

    def segments(self, images, cap=8192):
        """The raw segments of the detector core: list of B float32 [n_b, 4] (detection order)."""
        images = np.ascontiguousarray(images, np.uint8).reshape(self.B, self.rows, self.cols)
        seg = np.zeros((self.B, cap, 4), np.float32)
        n = np.zeros(self.B, np.int32)
        self.ctx._chk(self.ctx.lib.stvo_lsd_segments(self.h, images.reshape(-1), seg.reshape(-1), cap, n))
        return [seg[b, :min(n[b], cap)].copy() for b in range(self.B)], n


class Lbd:
    """The LBD line descriptor for B images of one size with up to M key-lines each (stvo_lbd_*)."""

    def __init__(self, ctx, B, cols, rows, max_keylines=512):
        self.ctx, self.B, self.cols, self.rows, self.M = ctx, B, cols, rows, max_keylines
        self.h = C.c_void_p()
        ctx._chk(ctx.lib.stvo_lbd_create(ctx.h, B, cols, rows, max_keylines, C.byref(self.h)))

    def close(self):
        if self.h:
            self.ctx.lib.stvo_lbd_destroy(self.h)
            self.h = None

    def compute_dev(self, img_ptr, lines_ptr, n_ptr, desc_ptr, desc_float_ptr=None):
        """Device pointers (uint8 [B, rows, cols]; stvo_keyline [B, M]; int32 [B]; uint8 [B, M, 32]; float32 [B, M, 72] or None)."""
        self.ctx._chk(self.ctx.lib.stvo_lbd_compute_dev(self.h, img_ptr, lines_ptr, n_ptr, desc_ptr, desc_float_ptr))

    def compute(self, images, lines, num_pixels, want_float=False):
        """images uint8 [B, rows, cols]; lines: list of B arrays [n_b, 5] (sx, sy, ex, ey, angle); num_pixels: list of B int arrays.
        -> list of B uint8 [n_b, 32] (and float32 [n_b, 72])."""
        images = np.ascontiguousarray(images, np.uint8).reshape(self.B, self.rows, self.cols)
        rec = np.zeros((self.B, self.M), KEYLINE_DTYPE)
        n = np.zeros(self.B, np.int32)
        for b in range(self.B):
            ln = np.asarray(lines[b], np.float32).reshape(-1, 5)
            n[b] = len(ln)
            for k, f in enumerate(("sx", "sy", "ex", "ey", "angle")):
                rec[f][b, :len(ln)] = ln[:, k]
            rec["num_pixels"][b, :len(ln)] = num_pixels[b]
        desc = np.zeros((self.B, self.M, 32), np.uint8)
        df = np.zeros((self.B, self.M, 72), np.float32) if want_float else None
        self.ctx._chk(self.ctx.lib.stvo_lbd_compute(self.h, images.reshape(-1), rec.ctypes.data_as(C.c_void_p), n, desc.reshape(-1),
                                                    df.ctypes.data_as(C.c_void_p) if want_float else None))
        out = [desc[b, :n[b]].copy() for b in range(self.B)]
        return (out, [df[b, :n[b]].copy() for b in range(self.B)]) if want_float else out


class Sequences:
    """B independent stereo sequences on the device-resident per-frame pipeline (stvo_seq_*)."""

    def __init__(self, ctx, B, max_kp, max_kl, cam, mp, op):
        """cam: one camera dict for all B sequences, or a list of B dicts (stvo_seq_create_multi)."""
        self.ctx, self.B = ctx, B
        self.h = C.c_void_p()
        if isinstance(cam, dict):
            camc = Cam.from_dict(cam)
            ctx._chk(ctx.lib.stvo_seq_create(ctx.h, B, max_kp, max_kl, cam["width"], cam["height"], C.byref(camc), C.byref(mp),
                                             C.byref(op), C.byref(self.h)))
        else:
            assert len(cam) == B
            cams = (Cam * B)(*[Cam.from_dict(c) for c in cam])
            cols = np.array([c["width"] for c in cam], np.int32)
            rows = np.array([c["height"] for c in cam], np.int32)
            ctx._chk(ctx.lib.stvo_seq_create_multi(ctx.h, B, max_kp, max_kl, cols, rows, C.cast(cams, C.c_void_p), C.byref(mp),
                                                   C.byref(op), C.byref(self.h)))

    def set_slots(self, n):
        self.ctx._chk(self.ctx.lib.stvo_seq_set_slots(self.h, n))

    def set_motion_model(self, on=True):
        """Config::useMotionModel(): the initial DT of a frame pair = the increment committed for the previous pair if it was good."""
        self.ctx._chk(self.ctx.lib.stvo_seq_set_motion_model(self.h, 1 if on else 0))

    def set_stage_timing(self, on=True):
        """True / 1: an event pair around every stage; 2: "light" — only the grid matcher, the forward scan and the pose kernel (the
        step otherwise as untimed); False: off."""
        self.ctx._chk(self.ctx.lib.stvo_seq_set_stage_timing(self.h, int(on)))

    def get_stage_timing(self):
        """({stage name: average ms per step}, number of steps measured) since the last call."""
        ms = (C.c_float * SEQ_NSTAGE)()
        n = C.c_int32()
        self.ctx._chk(self.ctx.lib.stvo_seq_get_stage_timing(self.h, ms, C.byref(n)))
        return {k: float(ms[i]) for i, k in enumerate(SEQ_STAGE_NAMES)}, n.value

    def debug_grid(self, b, lines):
        """Device-built grid of sequence b after the last step: (cell_start[3073], cell_items, cells_left, cand_off, cand)."""
        K, M = self.strides()
        cap_items = (M * 116) if lines else K
        cap_cand = (M * M) if lines else (K * 256)
        start = np.empty(64 * 48 + 1, np.int32); items = np.empty(cap_items, np.int32)
        cells = np.empty((M if lines else K) * (4 if lines else 2), np.int32)
        off = np.empty((M if lines else K) + 1, np.int32); cand = np.empty(cap_cand, np.int32)
        n = C.c_int32()
        self.ctx._chk(self.ctx.lib.stvo_seq_debug_grid(self.h, b, 1 if lines else 0, start, items, cap_items, cells, off, cand, cap_cand,
                                                      C.byref(n)))
        nl = n.value
        return (start, items[:start[-1]].copy(), cells[:nl * (4 if lines else 2)].reshape(nl, -1).copy(), off[:nl + 1].copy(),
                cand[:off[nl]].copy())

    def close(self):
        if self.h:
            self.ctx.lib.stvo_seq_destroy(self.h)
            self.h = None

    def _pack(self, frames):
        B = self.B
        assert len(frames) == B
        skp = max(max(len(f["kp_l"]), len(f["kp_r"])) for f in frames) or 1
        skl = max(max(len(f["kl_l"]), len(f["kl_r"])) for f in frames) or 1
        a = dict(n_kp_l=np.array([len(f["kp_l"]) for f in frames], np.int32), n_kp_r=np.array([len(f["kp_r"]) for f in frames], np.int32),
                 n_kl_l=np.array([len(f["kl_l"]) for f in frames], np.int32), n_kl_r=np.array([len(f["kl_r"]) for f in frames], np.int32),
                 kp_l=np.zeros((B, skp, 2), np.float32), oct_l=np.zeros((B, skp), np.int32), desc_l=np.zeros((B, skp, 32), np.uint8),
                 kp_r=np.zeros((B, skp, 2), np.float32), desc_r=np.zeros((B, skp, 32), np.uint8),
                 kl_l=np.zeros((B, skl, 4), np.float32), oct_ll=np.zeros((B, skl), np.int32), ldesc_l=np.zeros((B, skl, 32), np.uint8),
                 kl_r=np.zeros((B, skl, 4), np.float32), ldesc_r=np.zeros((B, skl, 32), np.uint8))
        for b, f in enumerate(frames):
            n = len(f["kp_l"]); a["kp_l"][b, :n] = f["kp_l"]; a["oct_l"][b, :n] = f["oct_l"]; a["desc_l"][b, :n] = f["desc_l"]
            n = len(f["kp_r"]); a["kp_r"][b, :n] = f["kp_r"]; a["desc_r"][b, :n] = f["desc_r"]
            n = len(f["kl_l"]); a["kl_l"][b, :n] = f["kl_l"]; a["oct_ll"][b, :n] = f["oct_ll"]; a["ldesc_l"][b, :n] = f["ldesc_l"]
            n = len(f["kl_r"]); a["kl_r"][b, :n] = f["kl_r"]; a["ldesc_r"][b, :n] = f["ldesc_r"]
        ff = FrameFeatures()
        ff.stride_kp, ff.stride_kl = skp, skl
        for k, v in a.items():
            setattr(ff, k, v.ctypes.data_as(C.c_void_p))
        return ff, a  # `a` keeps the arrays alive

    def push(self, frames):
        """frames: list of B per-sequence frame dicts as produced by synth.make_stereo_sequence."""
        ff, keep = self._pack(frames)
        res = np.zeros(self.B, dtype=POSE_RESULT_DTYPE)
        counts = np.zeros(self.B * 4, np.int32)
        self.ctx._chk(self.ctx.lib.stvo_seq_push(self.h, C.byref(ff), res.ctypes.data_as(C.c_void_p), counts))
        return res, counts.reshape(self.B, 4)

    def upload(self, slot, frames):
        ff, keep = self._pack(frames)
        self.ctx._chk(self.ctx.lib.stvo_seq_upload(self.h, slot, C.byref(ff)))
        self.ctx.synchronize()

    def upload_dev(self, slot, ff):
        """ff: FrameFeatures whose pointers (count arrays included) are DEVICE pointers; asynchronous."""
        self.ctx._chk(self.ctx.lib.stvo_seq_upload_dev(self.h, slot, C.byref(ff)))

    def step_dev(self, slot):
        self.ctx._chk(self.ctx.lib.stvo_seq_step_dev(self.h, slot))

    def read(self):
        res = np.zeros(self.B, dtype=POSE_RESULT_DTYPE)
        counts = np.zeros(self.B * 4, np.int32)
        self.ctx._chk(self.ctx.lib.stvo_seq_read(self.h, res.ctypes.data_as(C.c_void_p), counts))
        return res, counts.reshape(self.B, 4)

    def enable_fetch(self, on=True):
        self.ctx._chk(self.ctx.lib.stvo_seq_enable_fetch(self.h, 1 if on else 0))

    def strides(self):
        k, m = C.c_int32(), C.c_int32()
        self.ctx._chk(self.ctx.lib.stvo_seq_strides(self.h, C.byref(k), C.byref(m)))
        return k.value, m.value

    def fetch_matches(self):
        """(stereo m12 points [B,K], stereo m12 lines [B,M], f2f m12 points [B,K], f2f m12 lines [B,M]) of the last step."""
        K, M = self.strides()
        ps = [C.POINTER(C.c_int32)() for _ in range(4)]
        self.ctx._chk(self.ctx.lib.stvo_seq_fetch_matches(self.h, *[C.byref(p) for p in ps]))
        shapes = [(self.B, K), (self.B, M), (self.B, K), (self.B, M)]
        return tuple(np.ctypeslib.as_array(p, shape=sh).copy() for p, sh in zip(ps, shapes))

    def fetch_inliers(self):
        K, M = self.strides()
        ps = [C.POINTER(C.c_int32)() for _ in range(2)]
        self.ctx._chk(self.ctx.lib.stvo_seq_fetch_inliers(self.h, *[C.byref(p) for p in ps]))
        return tuple(np.ctypeslib.as_array(p, shape=sh).copy() for p, sh in zip(ps, [(self.B, K), (self.B, M)]))
