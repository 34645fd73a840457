"""The product's FP64 building blocks (stvo-pl_amd/csrc/pose_math.h — the code the HIP pose kernel
is assembled from) compiled for the HOST and checked against the oracle.  Runs without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import np_model
import oracle_lib
from stvo_amd import synth
from stvo_amd.ctypes_types import Cam, opt_params

HERE = os.path.dirname(os.path.abspath(__file__))
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def pmh():
    src = os.path.join(HERE, "cpp", "pm_host.cpp")
    so = os.path.join(HERE, "cpp", "libpm_host.so")
    hdr = os.path.join(HERE, "..", "stvo-pl_amd", "csrc", "pose_math.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src])
    lib = C.CDLL(so)
    for n in ("pmh_expmap", "pmh_logmap", "pmh_inverse_se3", "pmh_adjoint", "pmh_inverse6", "pmh_inverse6_mem", "pmh_eig6", "pmh_step_pose",
              "pmh_eig6_ql"):
        getattr(lib, n).argtypes = [f64p, f64p]; getattr(lib, n).restype = None
    lib.pmh_unccomp.argtypes = [f64p] * 4
    lib.pmh_solve6.argtypes = [f64p, f64p, f64p, C.POINTER(C.c_double)]; lib.pmh_solve6.restype = C.c_int
    lib.pmh_solve6_spd.argtypes = [f64p, f64p, f64p, C.POINTER(C.c_double)]; lib.pmh_solve6_spd.restype = C.c_int
    lib.pmh_solve6_mem.argtypes = [f64p, f64p, f64p, C.POINTER(C.c_double)]; lib.pmh_solve6_mem.restype = C.c_int
    lib.pmh_inverse6_spd.argtypes = [f64p, f64p]; lib.pmh_inverse6_spd.restype = C.c_int
    lib.pmh_line_overlap.argtypes = [f64p] * 4; lib.pmh_line_overlap.restype = C.c_double
    lib.pmh_normal_eq.argtypes = [f64p, C.POINTER(Cam), C.c_double, C.c_void_p, C.c_int, C.c_double, C.c_double, f64p]
    return lib


def call(lib, name, x, nout):
    out = np.empty(nout)
    getattr(lib, name)(np.ascontiguousarray(x, np.float64).reshape(-1), out)
    return out


def test_se3_blocks(pmh, oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        x = np.concatenate([rng.normal(0, 1, 3), rng.normal(0, 0.4, 3)])
        T = oracle.expmap(x)
        assert np.allclose(call(pmh, "pmh_expmap", x, 16).reshape(4, 4), T, atol=1e-15)
        assert np.allclose(call(pmh, "pmh_logmap", T, 6), oracle.logmap(T), atol=1e-13)
        assert np.allclose(call(pmh, "pmh_inverse_se3", T, 16).reshape(4, 4), oracle.inverse_se3(T), atol=1e-15)
        assert np.allclose(call(pmh, "pmh_adjoint", T, 36).reshape(6, 6), oracle.adjoint(T), atol=1e-15)
    x = np.array([1.0, 2.0, 3.0, 3e-7, 0, 0])
    T = call(pmh, "pmh_expmap", x, 16).reshape(4, 4)
    assert np.array_equal(T[:3, :3], np.eye(3)) and np.array_equal(T[:3, 3], x[:3])
    assert np.array_equal(call(pmh, "pmh_logmap", np.eye(4), 6), np.zeros(6))
    A = rng.normal(size=(6, 6)); c1 = A @ A.T; B = rng.normal(size=(6, 6)); ci = B @ B.T
    out = np.empty(36)
    pmh.pmh_unccomp(T.reshape(-1), c1.reshape(-1), ci.reshape(-1), out)
    assert np.allclose(out.reshape(6, 6), oracle.unccomp(T, c1, ci), rtol=1e-14)


def test_dense6_blocks(pmh, oracle):
    rng = np.random.default_rng(1)
    for k in range(200):
        J = rng.normal(size=(30, 6)) * np.array([1, 1, 1, 30, 30, 30.0])
        H = J.T @ J if k % 4 else rng.normal(size=(6, 6))
        g = rng.normal(size=6)
        x = np.empty(6); lad = C.c_double()
        rank = pmh.pmh_solve6(np.ascontiguousarray(H).reshape(-1), g, x, C.byref(lad))
        ox, olad, orank = oracle.solve6(H, g)
        assert rank == orank == 6
        assert np.allclose(x, ox, rtol=1e-12, atol=1e-14) and np.isclose(lad.value, olad, rtol=1e-13)
        assert np.allclose(call(pmh, "pmh_inverse6", H, 36), oracle.inverse6(H).reshape(-1), rtol=1e-11, atol=1e-15)
        assert np.allclose(call(pmh, "pmh_eig6", H, 6), oracle.eig6(H), rtol=1e-11, atol=1e-11)
    J = rng.normal(size=(3, 6)); H = J.T @ J; g = J.T @ np.ones(3)
    x = np.empty(6); lad = C.c_double()
    rank = pmh.pmh_solve6(H.reshape(-1).copy(), g, x, C.byref(lad))
    ox, _, orank = oracle.solve6(H, g)
    assert rank == orank == 3 and np.allclose(x, ox, atol=1e-9)
    # NaN input must terminate and propagate
    Hn = np.full((6, 6), np.nan)
    assert np.all(np.isnan(call(pmh, "pmh_eig6", Hn, 6)))
    pmh.pmh_solve6(Hn.reshape(-1).copy(), g, x, C.byref(lad))
    DT = np.eye(4).reshape(-1).copy()
    inc = rng.normal(0, 0.1, 6)
    pmh.pmh_step_pose(DT, inc)
    assert np.allclose(DT.reshape(4, 4), np.eye(4) @ np_model.inverse_se3(np_model.expmap_se3(inc)), atol=1e-15)


def test_memory_operand_forms_are_bit_identical(pmh):
    """pm::solve6_mem / pm::inverse6_mem (round 6: the pivoted fallbacks of the pose kernels on LDS operands with run-time loops, so that
    their 6 x 6 matrices do not cost the batch kernel registers) perform the operations of pm::solve6 / pm::inverse6 in the same order:
    identical bits, rank and log|det| on well-conditioned, ill-conditioned, rank-deficient, permuted, zero and NaN inputs."""
    rng = np.random.default_rng(21)
    mats = []
    for k in range(300):
        r = int(rng.integers(1, 7))
        J = rng.normal(size=(r if k % 3 == 0 else 30, 6)) * np.array([1, 1, 1, 30, 30, 30.0]) * 10.0 ** rng.uniform(-3, 3)
        H = J.T @ J
        if k % 5 == 0:
            H = rng.normal(size=(6, 6))                         # not symmetric: every pivot choice differs from the SPD case
        if k % 7 == 0:
            P = np.eye(6)[rng.permutation(6)]; H = P @ H @ P.T  # pivots away from the diagonal order
        mats.append(H)
    mats += [np.zeros((6, 6)), np.full((6, 6), np.nan), np.diag([1.0, 0, 2, 0, 3, 0]), np.ones((6, 6))]
    for H in mats:
        g = rng.normal(size=6)
        Hc = np.ascontiguousarray(H).reshape(-1)
        x0, x1, l0, l1 = np.empty(6), np.empty(6), C.c_double(), C.c_double()
        r0 = pmh.pmh_solve6(Hc.copy(), g.copy(), x0, C.byref(l0))
        r1 = pmh.pmh_solve6_mem(Hc.copy(), g.copy(), x1, C.byref(l1))
        assert r0 == r1 and x0.tobytes() == x1.tobytes() and np.array([l0.value]).tobytes() == np.array([l1.value]).tobytes()
        assert call(pmh, "pmh_inverse6", H, 36).tobytes() == call(pmh, "pmh_inverse6_mem", H, 36).tobytes()


def test_spd_fast_paths(pmh, oracle):
    """LDL^T solve / inverse and the tridiagonal-QL eigenvalues (what the kernel runs when H is certified SPD)
    against the pivoted routines of the oracle; tolerances scale with the condition number."""
    rng = np.random.default_rng(11)
    for k in range(300):
        scale = np.array([1, 1, 1, 30, 30, 30.0]) * 10.0 ** rng.uniform(-2, 3)
        J = rng.normal(size=(int(rng.integers(6, 60)), 6)) * scale
        H = J.T @ J
        g = J.T @ rng.normal(size=J.shape[0])
        cond = np.linalg.cond(H)
        x = np.empty(6); lad = C.c_double()
        ok = pmh.pmh_solve6_spd(np.ascontiguousarray(H).reshape(-1), g, x, C.byref(lad))
        ox, olad, _ = oracle.solve6(H, g)
        if cond < 1e9:
            assert ok == 1
        if ok:
            assert np.allclose(x, ox, rtol=1e-14 * cond + 1e-13, atol=1e-15 * cond * np.abs(ox).max())
            assert np.isclose(lad.value, olad, rtol=1e-12, atol=1e-10)
            Ai = np.empty(36)
            assert pmh.pmh_inverse6_spd(np.ascontiguousarray(H).reshape(-1), Ai) == 1
            oi = oracle.inverse6(H)
            assert np.allclose(Ai.reshape(6, 6), oi, rtol=1e-14 * cond + 1e-13, atol=1e-15 * cond * np.abs(oi).max())
            assert np.array_equal(Ai.reshape(6, 6), Ai.reshape(6, 6).T)
        C_ = np.linalg.inv(H)
        C_ = 0.5 * (C_ + C_.T)
        w = call(pmh, "pmh_eig6_ql", C_, 6)
        ref = np.linalg.eigvalsh(C_)
        assert np.allclose(w, ref, rtol=1e-9, atol=1e-14 * np.abs(ref).max())
        assert np.allclose(w, oracle.eig6(C_), rtol=1e-9, atol=1e-14 * np.abs(ref).max())
    # general symmetric (indefinite) matrices, repeated eigenvalues, diagonal and zero matrices
    for k in range(200):
        A = rng.normal(size=(6, 6)); A = A + A.T
        if k % 5 == 0:
            q, _ = np.linalg.qr(rng.normal(size=(6, 6))); A = q @ np.diag([1, 1, 1, 2, 2, -3.0]) @ q.T; A = 0.5 * (A + A.T)
        if k % 7 == 0:
            A = np.diag(rng.normal(size=6))
        w = call(pmh, "pmh_eig6_ql", A, 6)
        assert np.allclose(w, np.linalg.eigvalsh(A), rtol=1e-10, atol=1e-13)
    assert np.array_equal(call(pmh, "pmh_eig6_ql", np.zeros((6, 6)), 6), np.zeros(6))
    assert np.all(np.isnan(call(pmh, "pmh_eig6_ql", np.full((6, 6), np.nan), 6)))
    # not positive definite / rank deficient / NaN: the fast paths must decline
    x = np.empty(6); lad = C.c_double(); Ai = np.empty(36)
    J = rng.normal(size=(3, 6)); Hs = J.T @ J
    for bad in (Hs, -np.eye(6), np.zeros((6, 6)), np.full((6, 6), np.nan), np.diag([1, 1, 1, 1, 1, -1e-3])):
        b = np.ascontiguousarray(bad, np.float64).reshape(-1)
        assert pmh.pmh_solve6_spd(b, np.ones(6), x, C.byref(lad)) == 0
        assert pmh.pmh_inverse6_spd(b, Ai) == 0


def test_line_overlap_block(pmh, oracle):
    rng = np.random.default_rng(2)
    for _ in range(1000):
        so = rng.uniform(0, 400, 2); eo = so + rng.uniform(-200, 200, 2)
        k = rng.integers(0, 3)
        if k == 0: eo[0] = so[0] + rng.uniform(-0.99, 0.99)
        if k == 1: eo[1] = so[1] + rng.uniform(-0.99, 0.99)
        sp = so + rng.uniform(-150, 150, 2); ep = eo + rng.uniform(-150, 150, 2)
        a = pmh.pmh_line_overlap(so, eo, sp, ep); b = oracle.line_overlap(so, eo, sp, ep)
        assert a == b or (np.isnan(a) and np.isnan(b))


@pytest.mark.parametrize("robust", [0, 1])
def test_feature_terms_vs_oracle(pmh, oracle, robust):
    rec = synth.make_matched_records(7, n_pts=500, n_lines=80, octave_probs=[.5, .25, .15, .1])
    prm = opt_params("kitti")
    rec["inlier_p"][::9] = 0; rec["inlier_l"][::5] = 0
    m, keep = oracle_lib.Oracle._matched(rec)
    cam = Cam.from_dict(synth.KITTI_CAM)
    for DT in (np.eye(4), rec["T_true"]):
        s_p = s_l = 1.0
        if robust:
            clamp = lambda s: min(max(s, 1e-4), np.sqrt(7.815))
            s_p = clamp(oracle.stdv_mad(np_model.point_residuals(synth.KITTI_CAM, DT, rec)[rec["inlier_p"] > 0]))
            s_l = clamp(oracle.stdv_mad(np_model.line_residuals(synth.KITTI_CAM, DT, rec)[rec["inlier_l"] > 0]))
        acc = np.empty(28)
        pmh.pmh_normal_eq(np.ascontiguousarray(DT).reshape(-1), C.byref(cam), prm.homog_th, C.addressof(m), robust, s_p, s_l, acc)
        H, g, e, n = oracle.optimize_functions(DT, synth.KITTI_CAM, prm, rec, robust)
        Hd = np.zeros((6, 6)); k = 0
        for i in range(6):
            for j in range(i, 6):
                Hd[i, j] = Hd[j, i] = acc[k]; k += 1
        assert np.allclose(Hd, H, rtol=1e-12, atol=1e-12 * np.abs(H).max())
        assert np.allclose(acc[21:27], g, rtol=1e-12, atol=1e-12 * np.abs(g).max())
        assert np.isclose(acc[27] / n, e, rtol=1e-13)


def test_spd_certificate_never_contradicts_eig(pmh):
    """spd_unit_certificate (the pose kernel's shortcut for isGoodSolution's eigenvalue test) may answer
    'undecided' but must never certify a matrix whose eigenvalues violate 0 <= lambda <= 1."""
    pmh.pmh_spd_cert.argtypes = [f64p]; pmh.pmh_spd_cert.restype = C.c_int
    rng = np.random.default_rng(5)
    n_cert = 0
    for k in range(3000):
        kind = k % 6
        A = rng.normal(size=(6, 6))
        if kind == 0: M = A @ A.T * 10 ** rng.uniform(-9, -1)          # typical covariances: tiny SPD
        elif kind == 1: M = A @ A.T * 10 ** rng.uniform(-1, 1)         # around the lambda_max = 1 boundary
        elif kind == 2: M = (A + A.T) * 0.01                            # indefinite
        elif kind == 3: J = rng.normal(size=(4, 6)); M = J.T @ J * 0.01  # singular PSD
        elif kind == 4: M = np.diag(rng.uniform(-0.1, 1.2, 6))
        else: M = A @ A.T * 1e-6; M[0, 0] = np.nan if k % 12 == 5 else M[0, 0]
        c = pmh.pmh_spd_cert(np.ascontiguousarray(M).reshape(-1))
        if c == 1:
            n_cert += 1
            L = np.tril(M); S = L + L.T - np.diag(np.diag(M))
            w = np.linalg.eigvalsh(S)
            assert w[0] >= 0.0 and w[-1] <= 1.0, (kind, w)
    assert n_cert > 500  # the shortcut actually fires on the common case


def test_keyframe_decision(pmh):
    """needNewKF / currFrameIsKF (src/stereoFrameHandler.cpp:1134-1218) of the host mirror against a numpy model:
    entropy of the first frame after a key-frame, accumulated covariance through Ad(T_prevKF) uncTinv(DT, DT_cov)
    Ad^T, geometric thresholds, the 10-frame limit and the DT == I / cov == 0 (failed frame) rule."""
    i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
    pmh.pmh_kf_sequence.argtypes = [C.c_int, f64p, f64p, f64p, C.c_double, C.c_double, C.c_double, i32p, f64p, f64p]
    pmh.pmh_det6.argtypes = [f64p]; pmh.pmh_det6.restype = C.c_double
    pmh.pmh_unctinv.argtypes = [f64p, f64p, f64p]
    rng = np.random.default_rng(5)
    for _ in range(50):
        A = rng.normal(size=(6, 6))
        assert np.isclose(pmh.pmh_det6(np.ascontiguousarray(A).reshape(-1)), np.linalg.det(A), rtol=1e-11, atol=1e-13)
    assert pmh.pmh_det6(np.zeros(36)) == 0.0

    def adj(T):
        R, t = T[:3, :3], T[:3, 3]
        sk = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        Ad = np.zeros((6, 6)); Ad[:3, :3] = R; Ad[:3, 3:] = sk @ R; Ad[3:, 3:] = R
        return Ad

    n = 60
    Tfw = np.zeros((n, 4, 4)); DT = np.zeros((n, 4, 4)); DC = np.zeros((n, 6, 6))
    T = np.eye(4)
    for f in range(n):
        inc = np.concatenate([rng.normal(0, 0.02, 2), [-rng.uniform(0.3, 1.2)], rng.normal(0, 0.01, 3)])
        D = np_model.expmap_se3(inc)
        J = rng.normal(size=(20, 6)) * np.array([30, 30, 30, 300, 300, 300.0]) * rng.uniform(0.5, 3)
        Cv = np.linalg.inv(J.T @ J)
        if f % 17 == 9:  # a failed frame as optimizePose leaves it (:382-391)
            D = np.eye(4); Cv = np.zeros((6, 6))
        T = T @ D
        Tfw[f], DT[f], DC[f] = T, D, Cv
    # numpy model (Tfw is taken relative to the last key-frame exactly as the C side is driven: state reset to identity)
    exp_out = np.zeros(n, np.int32); exp_acc = np.zeros((n, 6, 6)); exp_ent = np.zeros(n)
    k_is, k_ent, k_T, k_cov, k_N = True, 0.0, np.eye(4), np.zeros((6, 6)), 0
    c0 = 3.0 * (1.0 + np.log(2.0 * np.pi))
    for f in range(n):
        if k_is:
            det = np.linalg.det(DC[f])
            k_ent = c0 + 0.5 * np.log(det) if det != 0.0 else -999999999.99
            k_is = False
        dX = np_model.logmap_se3(np_model.inverse_se3(Tfw[f]) @ k_T)
        t = np.linalg.norm(dX[:3]); r = np.linalg.norm(dX[3:]) * 180.0 / np.pi
        Ai = adj(np_model.inverse_se3(DT[f])); cinv = Ai @ DC[f] @ Ai.T
        Ak = adj(k_T); k_cov = k_cov + Ak @ cinv @ Ak.T
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = (c0 + 0.5 * np.log(np.linalg.det(k_cov))) / k_ent
        need = bool(ratio < 0.85 or np.isnan(ratio) or np.isinf(ratio) or (not DC[f].any() and np.array_equal(DT[f], np.eye(4)))
                    or t > 5.0 or r > 15.0 or k_N > 10)
        exp_out[f] = need; exp_acc[f] = k_cov; exp_ent[f] = k_ent
        if need:
            k_T, k_cov, k_is, k_N = np.eye(4), np.zeros((6, 6)), True, 0
        else:
            k_N += 1
    out = np.zeros(n, np.int32); acc = np.zeros((n, 36)); ent = np.zeros(n)
    pmh.pmh_kf_sequence(n, Tfw.reshape(-1).copy(), DT.reshape(-1).copy(), DC.reshape(-1).copy(), 0.85, 5.0, 15.0, out, acc.reshape(-1), ent)
    assert np.array_equal(out, exp_out)
    assert 3 <= out.sum() < n  # both outcomes occur
    assert np.allclose(acc.reshape(n, 6, 6), exp_acc, rtol=1e-9, atol=1e-18)
    assert np.allclose(ent, exp_ent, rtol=1e-10)
    # the oracle's C restatement (oracle/stvo_oracle.c: orc_need_new_kf / orc_curr_frame_is_kf) takes the same decisions and
    # accumulates the same covariance: a third, independent statement of the rule
    import oracle_lib
    orc = oracle_lib.load()
    st = orc.kf_state()
    for f in range(n):
        need = orc.need_new_kf(st, Tfw[f], DT[f], DC[f])
        assert need == exp_out[f], f
        assert np.allclose(st[19:55].reshape(6, 6), exp_acc[f], rtol=1e-9, atol=1e-18)
        if need:
            orc.curr_frame_is_kf(st)
