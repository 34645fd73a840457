"""GPU parity: the pose kernel (optimizeFunctions / optimizePose on device) through the C-ABI vs the
oracle.  Tolerance: the north star's 1e-4 rad / 1e-3 m per frame on the pose; in practice the two
agree to ~1e-9 because both are FP64 and differ only in summation order.  Inlier masks, iteration
counts and the path through the state machine must be identical."""
import numpy as np
import pytest

import np_model
from stvo_amd import synth
from stvo_amd.ctypes_types import opt_params

pytestmark = pytest.mark.gpu
CAM = synth.KITTI_CAM


@pytest.fixture(autouse=True, params=["default", "1", "4:2", "4:4"])
def pose_kernel_variant(request, switches):
    """Every test of this module runs against both optimizePose formulations of the library: pose_kernel.hip (worker waves + a
    solver wave, every record in LDS: "1", the default up to 256 frame pairs) and pose_kernel2p.hip (every wave a worker,
    thread-private records, two or four waves per frame pair: "4:2" / "4:4", the default for larger batches); "default" is the
    library's own choice.  (The round-2 kernel with compacted LDS records and the round-3 owner / evaluator experiment were
    removed in round 4 — both measured slower.)"""
    if request.param != "default":
        k, _, nw = request.param.partition(":")
        switches(dict({"STVO_POSE_KERNEL": k}, **({"STVO_POSE2P_NW": nw} if nw else {})))
    yield request.param


ROT_TOL, TRANS_TOL = 1e-4, 1e-3  # BASELINE.json north_star


@pytest.mark.parametrize("npts,nl,robust", [(300, 0, 0), (300, 40, 0), (0, 60, 0), (250, 30, 1), (1500, 80, 0),
                                             (2048, 512, 0), (2048, 512, 1), (1, 0, 0), (0, 1, 0)])
def test_normal_eq_vs_oracle(hip, oracle, npts, nl, robust):
    rec = synth.make_matched_records(100 + npts + nl, n_pts=npts, n_lines=nl, octave_probs=[.5, .25, .15, .1])
    prm = opt_params("kitti")
    rng = np.random.default_rng(npts + 3)
    if npts:
        rec["inlier_p"][rng.random(npts) < 0.1] = 0
    for DT in (np.eye(4), rec["T_true"]):
        H, g, e, n = hip.normal_eq(DT, CAM, prm, rec, robust)
        oH, og, oe, on = oracle.optimize_functions(DT, CAM, prm, rec, robust)
        scale = max(np.abs(oH).max(), 1e-300)
        assert np.allclose(H, oH, rtol=1e-10, atol=1e-11 * scale)
        assert np.allclose(g, og, rtol=1e-10, atol=1e-11 * max(np.abs(og).max(), 1e-300))
        assert (np.isnan(e) and np.isnan(oe)) or np.isclose(e, oe, rtol=1e-11)
        assert n == on
        assert np.array_equal(H, H.T)


def check_pose(out, ref):
    assert out["status"] == ref["status"] and out["path"] == ref["path"], (out["status"], ref["status"], out["path"], ref["path"])
    assert out["iters"] == ref["iters"]
    assert np.array_equal(out["inlier_p"], ref["inlier_p"]) and np.array_equal(out["inlier_l"], ref["inlier_l"])
    assert out["n_inliers_pt"] == ref["n_inliers_pt"] and out["n_inliers_ls"] == ref["n_inliers_ls"]
    assert np_model.rot_angle(out["T"][:3, :3], ref["T"][:3, :3]) < ROT_TOL
    assert np.linalg.norm(out["T"][:3, 3] - ref["T"][:3, 3]) < TRANS_TOL
    # far tighter in practice
    assert np.allclose(out["T"], ref["T"], atol=1e-8)
    assert np.allclose(out["cov"], ref["cov"], rtol=1e-6, atol=1e-12)
    assert np.allclose(out["cov_eig"], ref["cov_eig"], rtol=1e-6, atol=1e-13)
    assert np.isclose(out["err"], ref["err"], rtol=1e-8)


@pytest.mark.parametrize("seed,npts,nl,mode,preset", [(31, 1200, 0, 0, "kitti"), (32, 1000, 60, 0, "kitti"),
                                                      (33, 600, 200, 0, "euroc"), (34, 600, 200, 2, "euroc"),
                                                      (35, 600, 200, 1, "euroc"), (36, 60, 5, 0, "kitti"),
                                                      (37, 2048, 512, 0, "kitti"), (38, 1400, 0, 0, "kitti"),
                                                      (39, 0, 300, 0, "euroc"), (40, 800, 300, 2, "kitti")])
def test_optimize_pose_vs_oracle(hip, oracle, seed, npts, nl, mode, preset):
    rec = synth.make_matched_records(seed, n_pts=npts, n_lines=nl,
                                     octave_probs=[.5, .25, .15, .1] if preset == "euroc" else None,
                                     outlier_frac=0.4 if preset == "euroc" else 0.15)
    prm = opt_params(preset, mode=mode)
    out = hip.optimize_pose(np.eye(4), CAM, prm, rec)
    ref = oracle.optimize_pose(np.eye(4), CAM, prm, rec)
    check_pose(out, ref)


def test_failure_paths(hip, oracle):
    prm = opt_params("kitti")
    for seed, npts, frac in ((41, 8, 0.15), (42, 200, 1.0), (43, 12, 0.5)):
        rec = synth.make_matched_records(seed, n_pts=npts, n_lines=0, outlier_frac=frac)
        out = hip.optimize_pose(np.eye(4), CAM, prm, rec)
        ref = oracle.optimize_pose(np.eye(4), CAM, prm, rec)
        check_pose(out, ref)
    rec = synth.make_matched_records(41, n_pts=8, n_lines=0)
    out = hip.optimize_pose(np.eye(4), CAM, prm, rec)
    assert out["status"] == 1 and np.array_equal(out["T"], np.eye(4)) and out["err"] == -1.0 and not out["cov"].any()
    # empty problem
    rec = synth.make_matched_records(44, n_pts=0, n_lines=0)
    out = hip.optimize_pose(np.eye(4), CAM, prm, rec)
    assert out["status"] == 1


def test_zero_noise_recovers_motion(hip):
    rec = synth.make_matched_records(21, n_pts=400, n_lines=50, outlier_frac=0.0, noise_px=0.0)
    prm = opt_params("kitti", max_iters=30, max_iters_ref=30)
    out = hip.optimize_pose(np.eye(4), CAM, prm, rec)
    assert out["status"] == 0
    assert np.allclose(out["T_opt"], rec["T_true"], atol=1e-5)


def test_init_T_and_masks(hip, oracle):
    rec = synth.make_matched_records(55, n_pts=900, n_lines=100)
    rec["inlier_p"][::5] = 0
    rec["inlier_l"][::3] = 0
    init = np_model.expmap_se3(np.array([0.01, -0.02, -0.5, 0.002, -0.001, 0.003]))
    prm = opt_params("kitti")
    check_pose(hip.optimize_pose(init, CAM, prm, rec), oracle.optimize_pose(init, CAM, prm, rec))
    prm = opt_params("kitti", has_lines=0)
    check_pose(hip.optimize_pose(init, CAM, prm, rec), oracle.optimize_pose(init, CAM, prm, rec))


def test_deterministic(hip):
    rec = synth.make_matched_records(61, n_pts=1500, n_lines=80)
    prm = opt_params("kitti")
    a = hip.optimize_pose(np.eye(4), CAM, prm, rec)
    b = hip.optimize_pose(np.eye(4), CAM, prm, rec)
    assert np.array_equal(a["T"], b["T"]) and np.array_equal(a["cov"], b["cov"]) and a["err"] == b["err"]


def test_rank_deficient_systems_take_the_pivoted_path(hip, oracle):
    """With fewer than six independent residuals H = sum w J J^T is singular: the LDL^T fast path of the solver wave
    must decline and the column-pivoted QR / pivoted LU restatements of Eigen's routines run instead, as in the oracle.
    Degenerate problems amplify last-bit differences (the device contracts a*b+c into FMAs, the oracle does not), so
    iteration counts and poses are only required to agree where the problem still has a stable answer (3 and 4
    points); status and state-machine path must agree everywhere, and nothing may be non-finite without the
    reference semantics asking for it."""
    for seed, npts, nl, strict in ((71, 3, 0, True), (72, 4, 0, True), (73, 5, 0, False), (74, 0, 4, False), (75, 2, 2, False)):
        rec = synth.make_matched_records(seed, n_pts=npts, n_lines=nl, outlier_frac=0.0)
        prm = opt_params("kitti", min_features=2)
        out = hip.optimize_pose(np.eye(4), CAM, prm, rec)
        ref = oracle.optimize_pose(np.eye(4), CAM, prm, rec)
        assert out["status"] == ref["status"] and out["path"] == ref["path"], (seed, out["status"], ref["status"], out["path"], ref["path"])
        assert np.all(np.isfinite(out["T"])) and np.all(np.isfinite(out["cov"]))
        if strict:
            assert out["iters"] == ref["iters"], (seed, out["iters"], ref["iters"])
            assert np.array_equal(out["inlier_p"], ref["inlier_p"]) and np.array_equal(out["inlier_l"], ref["inlier_l"])
            assert np.allclose(out["T"], ref["T"], atol=1e-6), seed
            assert np.allclose(out["T_opt"], ref["T_opt"], atol=1e-6, equal_nan=True), seed
