"""Independent numpy float64 model of the hot path — TEST INFRASTRUCTURE ONLY.

A second restatement, written separately from oracle/stvo_oracle.c (vectorised numpy,
numpy.linalg for the 6x6 algebra), used to cross-check the C oracle because the reference holds no
golden vectors for this path.  Cites the same reference lines.
"""
import numpy as np

_POP = np.array([bin(i).count("1") for i in range(256)], dtype=np.int32)


def hamming_matrix(d1, d2):
    """All-pairs Hamming distances of N1x32 and N2x32 uint8 descriptor sets."""
    x = d1[:, None, :] ^ d2[None, :, :]
    return _POP[x].sum(axis=2).astype(np.int32)


def knn2(D):
    """Two smallest per row, lowest index first among ties (stable argsort)."""
    order = np.argsort(D, axis=1, kind="stable")
    i0 = order[:, 0]
    d0 = D[np.arange(len(D)), i0]
    d1 = D[np.arange(len(D)), order[:, 1]]
    return i0.astype(np.int32), d0, d1


def match_nnr_from_D(D, nnr):
    """src/matching.cpp:41-61 with the FLOAT ratio test."""
    if D.shape[0] == 0:
        return np.zeros(0, np.int32)
    if D.shape[1] < 2:
        return -np.ones(D.shape[0], np.int32)
    i0, d0, d1 = knn2(D)
    ok = d0.astype(np.float32) < d1.astype(np.float32) * np.float32(nnr)
    return np.where(ok, i0, -1).astype(np.int32)


def match(d1, d2, nnr, best_lr=True):
    """src/matching.cpp:63-91"""
    D = hamming_matrix(d1, d2)
    m12 = match_nnr_from_D(D, nnr)
    if best_lr:
        m21 = match_nnr_from_D(D.T.copy(), nnr)
        for i1 in range(len(m12)):
            i2 = m12[i1]
            if i2 >= 0 and m21[i2] != i1:
                m12[i1] = -1
    return m12


def match_grid(cand_lists, d1, d2, ratio, best_lr=True, gate=None):
    """src/matching.cpp:111-177 / :179-258 from explicit candidate lists, visited in REVERSED order
    (the unordered_set iteration order is implementation-defined and must not matter)."""
    n1, n2 = len(d1), len(d2)
    INT_MAX = 2**31 - 1
    dist = np.full(n2, INT_MAX, np.int64)
    m21 = -np.ones(n2, np.int64)
    m12 = -np.ones(n1, np.int32)
    for i1 in range(n1):
        best_d = best_d2 = INT_MAX
        best = -1
        for i2 in reversed(list(cand_lists[i1])):
            if gate is not None and not gate(i1, i2):
                continue
            d = int(_POP[d1[i1] ^ d2[i2]].sum())
            if best_lr:
                if d < dist[i2]:
                    dist[i2] = d
                    m21[i2] = i1
                else:
                    continue
            if d < best_d:
                best_d2, best_d, best = best_d, d, i2
            elif d < best_d2:
                best_d2 = d
        if float(best_d) < float(best_d2) * ratio:
            m12[i1] = best
    if best_lr:
        for i1 in range(n1):
            i2 = m12[i1]
            if i2 >= 0 and m21[i2] != i1:
                m12[i1] = -1
    return m12


# ------------------------------------------------------------------------------------------------
def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def expmap_se3(x):
    """src/auxiliar.cpp:124-141"""
    t, w = np.array(x[:3], float), np.array(x[3:], float)
    th = np.linalg.norm(w)
    T = np.eye(4)
    if th < 1e-6:
        R = np.eye(3)
    else:
        s = skew(w) / th
        R = np.eye(3) + s * np.sin(th) + s @ s * (1 - np.cos(th))
        V = np.eye(3) + s * (1 - np.cos(th)) / th + s @ s * (th - np.sin(th)) / th
        t = V @ t
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def logmap_se3(T):
    """src/auxiliar.cpp:143-173"""
    R = T[:3, :3]
    c = min(1.0, max(-1.0, (np.trace(R) - 1) / 2))
    s = np.sqrt(1 - c * c)
    th = np.arccos(c)
    V = np.eye(3)
    w = np.zeros(3)
    if th > 1e-6:
        wh = th * (R - R.T) / (2 * s)
        w = np.array([wh[2, 1], wh[0, 2], wh[1, 0]])
        k = skew(w) / th
        V = np.eye(3) + k * (1 - c) / th + k @ k * (th - s) / th
    return np.concatenate([np.linalg.solve(V, T[:3, 3]), w])


def inverse_se3(T):
    Ti = np.eye(4)
    Ti[:3, :3] = T[:3, :3].T
    Ti[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return Ti


def adjoint_se3(T):
    A = np.zeros((6, 6))
    R = T[:3, :3]
    A[:3, :3] = R
    A[:3, 3:] = skew(T[:3, 3]) @ R
    A[3:, 3:] = R
    return A


def mean_stdv_mad(r):
    """src/auxiliar.cpp:387-430"""
    r = np.asarray(r, float)
    n = len(r)
    if n == 0:
        return 0.0, 0.0
    s = np.sort(r)
    med = s[n // 2]
    dev = np.sort(np.abs((s - med).astype(np.float32)).astype(np.float64))
    stdv = 1.4826 * dev[n // 2]
    sel = r < 2.0 * stdv
    k = int(sel.sum())
    if k >= int(0.2 * n):
        with np.errstate(invalid="ignore", divide="ignore"):
            mean = np.float64(sum(r[sel].tolist())) / np.float64(k)
    else:
        mean = sum(r.tolist()) / n
    return float(mean), float(stdv)


def stdv_mad(r):
    r = np.asarray(r, float)
    n = len(r)
    if n == 0:
        return 0.0
    s = np.sort(r)
    dev = np.sort(np.abs((s - s[n // 2]).astype(np.float32)).astype(np.float64))
    return 1.4826 * dev[n // 2]


def _lambdas(ls, le):
    lmin, lmax = min(ls, le), max(ls, le)
    if lmin < 0 and lmax > 1:
        return 1.0
    if lmax < 0 or lmin > 1:
        return 0.0
    if lmin < 0:
        return lmax
    if lmax > 1:
        return 1.0 - lmin
    return lmax - lmin


def line_overlap(so, eo, sp, ep):
    """src/stereoFrame.cpp:510-616"""
    l = eo - so
    with np.errstate(all="ignore"):
        if abs(so[0] - eo[0]) < 1.0:
            return _lambdas((sp[1] - so[1]) / l[1], (ep[1] - so[1]) / l[1])
        if abs(so[1] - eo[1]) < 1.0:
            return _lambdas((sp[0] - so[0]) / l[0], (ep[0] - so[0]) / l[0])
        a = so[1] - eo[1]; b = eo[0] - so[0]; c = so[0] * eo[1] - eo[0] * so[1]
        lxy = 1.0 / (a * a + b * b)
        spx = (b * (b * sp[0] - a * sp[1]) - a * c) * lxy
        epx = (b * (b * ep[0] - a * ep[1]) - a * c) * lxy
        return _lambdas((spx - so[0]) / l[0], (epx - so[0]) / l[0])


def _proj(cam, DT, P):
    Pc = P @ DT[:3, :3].T + DT[:3, 3]
    uv = np.stack([cam["cx"] + cam["fx"] * Pc[:, 0] / Pc[:, 2], cam["cy"] + cam["fy"] * Pc[:, 1] / Pc[:, 2]], 1)
    return Pc, uv


def _grad(Pc, dx, dy, fx, homog_th):
    gx, gy, gz = Pc[:, 0], Pc[:, 1], Pc[:, 2]
    f = fx / np.maximum(homog_th, gz * gz)
    return np.stack([f * dx * gz, f * dy * gz, -f * (gx * dx + gy * dy), -f * (gx * gy * dx + gy * gy * dy + gz * gz * dy),
                     f * (gx * gx * dx + gz * gz * dx + gx * gy * dy), f * (gx * gz * dy - gy * gz * dx)], 1)


def point_residuals(cam, DT, rec):
    _, uv = _proj(cam, DT, rec["P"])
    return np.linalg.norm(uv - rec["pl_obs"], axis=1)


def line_residuals(cam, DT, rec):
    _, s = _proj(cam, DT, rec["sP"]); _, t = _proj(cam, DT, rec["eP"])
    l = rec["le_obs"]
    ds = l[:, 0] * s[:, 0] + l[:, 1] * s[:, 1] + l[:, 2]
    de = l[:, 0] * t[:, 0] + l[:, 1] * t[:, 1] + l[:, 2]
    return np.sqrt(ds * ds + de * de)


def optimize_functions(DT, cam, prm, rec, inl_p, inl_l, robust=False):
    """src/stereoFrameHandler.cpp:549-694 (robust: :696-962)"""
    H = np.zeros((6, 6)); g = np.zeros(6); e = 0.0
    hth = prm["homog_th"]
    ip = np.asarray(inl_p, bool); il = np.asarray(inl_l, bool)
    s_p = s_l = 1.0
    if robust:
        clamp = lambda s: min(max(s, 1e-4), np.sqrt(7.815))
        s_p = clamp(stdv_mad(point_residuals(cam, DT, rec)[ip])) if len(ip) else clamp(0.0)
        s_l = clamp(stdv_mad(line_residuals(cam, DT, rec)[il])) if len(il) else clamp(0.0)
    n = 0
    if ip.any():
        Pc, uv = _proj(cam, DT, rec["P"][ip])
        d = uv - rec["pl_obs"][ip]
        nrm = np.linalg.norm(d, axis=1)
        J = _grad(Pc, d[:, 0], d[:, 1], cam["fx"], hth) / np.maximum(hth, nrm)[:, None]
        if robust:
            r = nrm; w = 1 / (1 + (r / s_p) ** 2)
        else:
            r = nrm * np.sqrt(rec["sigma2p"][ip]); w = 1 / (1 + r * r)
        H += (J * w[:, None]).T @ J; g += J.T @ (r * w); e += float(np.sum(r * r * w)); n += int(ip.sum())
    if il.any():
        sPc, s = _proj(cam, DT, rec["sP"][il]); ePc, t = _proj(cam, DT, rec["eP"][il])
        l = rec["le_obs"][il]
        ds = l[:, 0] * s[:, 0] + l[:, 1] * s[:, 1] + l[:, 2]
        de = l[:, 0] * t[:, 0] + l[:, 1] * t[:, 1] + l[:, 2]
        nrm = np.sqrt(ds * ds + de * de)
        Js = _grad(sPc, l[:, 0], l[:, 1], cam["fx"], hth); Je = _grad(ePc, l[:, 0], l[:, 1], cam["fx"], hth)
        J = (Js * ds[:, None] + Je * de[:, None]) / np.maximum(hth, nrm)[:, None]
        if robust:
            r = nrm; w = 1 / (1 + (r / s_l) ** 2)
        else:
            r = nrm * np.sqrt(rec["sigma2l"][il]); w = 1 / (1 + r * r)
        ov = np.array([line_overlap(a, b, c, d_) for a, b, c, d_ in zip(rec["spl"][il], rec["epl"][il], s, t)])
        w = w * ov
        H += (J * w[:, None]).T @ J; g += J.T @ (r * w); e += float(np.sum(r * r * w)); n += int(il.sum())
    with np.errstate(all="ignore"):
        e = np.float64(e) / np.float64(n)
    return H, g, float(e)


def _step(DT, inc):
    return DT @ inverse_se3(expmap_se3(inc))


def gauss_newton(DT, cam, prm, rec, ip, il, max_iters):
    """src/stereoFrameHandler.cpp:394-431; returns DT, cov, err, evals"""
    err_prev = 999999999.9
    H = np.zeros((6, 6)); err = 0.0; evals = 0
    for it in range(max_iters):
        H, g, err = optimize_functions(DT, cam, prm, rec, ip, il)
        evals += 1
        if err > err_prev:
            if it > 0:
                break
            return DT, None, -1.0, evals
        if err < prm["min_error"] or abs(err - err_prev) < prm["min_error_change"]:
            break
        inc = np.linalg.solve(H, g)
        DT = _step(DT, inc)
        if np.linalg.norm(inc[:3]) < prm["min_error_change"] and np.linalg.norm(inc[3:]) < prm["min_error_change"]:
            break
        err_prev = err
    return DT, np.linalg.inv(H), err, evals


def gauss_newton_robust(DT, cam, prm, rec, ip, il, max_iters):
    """:433-480"""
    DT0 = DT.copy(); err_prev = 999999999.9; good = True; evals = 0
    H = np.zeros((6, 6)); err = 0.0
    for it in range(max_iters):
        H, g, err = optimize_functions(DT, cam, prm, rec, ip, il, robust=True)
        evals += 1
        if abs(err - err_prev) < prm["min_error_change"] or err < prm["min_error"]:
            break
        inc = np.linalg.solve(H, g)
        if np.linalg.slogdet(H)[1] < 0:
            good = False
            break
        DT = _step(DT, inc)
        if np.linalg.norm(inc) < prm["min_error_change"]:
            break
        err_prev = err
    if good:
        return DT, np.linalg.inv(H), err, evals
    return DT0, np.eye(6), -1.0, evals


def levenberg_marquardt(DT, cam, prm, rec, ip, il, max_iters):
    """:482-547"""
    lam = 1e-9; k = 4.0
    H, g, err = optimize_functions(DT, cam, prm, rec, ip, il); evals = 1
    lam *= np.max(np.abs(np.diag(H)))
    H = H + lam * np.eye(6)
    DT = _step(DT, np.linalg.solve(H, g))
    err_prev = err
    for it in range(1, max_iters):
        H, g, err = optimize_functions(DT, cam, prm, rec, ip, il); evals += 1
        if abs(err - err_prev) < prm["min_error_change"] or err < prm["min_error"]:
            break
        H = H + lam * np.eye(6)
        inc = np.linalg.solve(H, g)
        if err > err_prev:
            lam /= k
        else:
            lam *= k
            DT = _step(DT, inc)
        if np.linalg.norm(inc[:3]) < prm["min_error_change"] and np.linalg.norm(inc[3:]) < prm["min_error_change"]:
            break
        err_prev = err
    return DT, np.linalg.inv(H), err, evals


def is_good(DT, cov, err):
    """:292-305"""
    if cov is None:
        return False
    with np.errstate(all="ignore"):
        L = np.tril(cov); S = L + L.T - np.diag(np.diag(cov))
        try:
            w = np.linalg.eigvalsh(S)
        except np.linalg.LinAlgError:
            return False
    return not (w[0] < 0 or w[5] > 1 or err < 0 or err > 1 or not np.all(np.isfinite(DT)))


def remove_outliers(DT, cam, prm, rec, ip, il):
    """:988-1067"""
    ip = np.array(ip, bool); il = np.array(il, bool)
    if prm["has_points"]:
        res = point_residuals(cam, DT, rec) * np.sqrt(rec["sigma2p"]) if len(ip) else np.zeros(0)
        mean, stdv = mean_stdv_mad(res)
        ip &= ~(np.abs(res - mean) > prm["inlier_k"] * stdv)
    if prm["has_lines"]:
        res = line_residuals(cam, DT, rec) * np.sqrt(rec["sigma2l"]) if len(il) else np.zeros(0)
        mean, stdv = mean_stdv_mad(res)
        il &= ~(np.abs(res - mean) > prm["inlier_k"] * stdv)
    return ip, il


def optimize_pose(init_T, cam, prm, rec):
    """:307-392 (mode from prm['mode'])"""
    run = {0: gauss_newton, 1: gauss_newton_robust, 2: levenberg_marquardt}[prm["mode"]]
    ip = np.array(rec["inlier_p"], bool); il = np.array(rec["inlier_l"], bool)
    DT = np.array(init_T, float); cov = None; err = -1.0
    status = 0; path = 0; iters = [0, 0]
    if ip.sum() + il.sum() >= prm["min_features"]:
        DT_, cov1, err1, iters[0] = run(DT.copy(), cam, prm, rec, ip, il, prm["max_iters"])
        cov, err = cov1, err1
        if is_good(DT_, cov1, err1):
            path |= 1
            ip, il = remove_outliers(DT_, cam, prm, rec, ip, il)
            if ip.sum() + il.sum() >= prm["min_features"]:
                path |= 4
                DT2, cov2, err, iters[1] = run(DT.copy(), cam, prm, rec, ip, il, prm["max_iters_ref"])
                DT = DT2
                cov = cov2 if cov2 is not None else cov1
            else:
                DT = np.eye(4); status = 2
        else:
            path |= 2
            DT, cov, err, iters[1] = gauss_newton_robust(DT.copy(), cam, prm, rec, ip, il, prm["max_iters_ref"])
    else:
        DT = np.eye(4); status = 1
    out = dict(T_opt=DT, err_opt=err, path=path, iters=tuple(iters), inlier_p=ip.astype(np.int32), inlier_l=il.astype(np.int32))
    if is_good(DT, cov, err) and not np.array_equal(DT, np.eye(4)):
        out.update(T=expmap_se3(logmap_se3(inverse_se3(DT))), cov=cov, err=err, status=status)
        L = np.tril(cov)
        out["cov_eig"] = np.linalg.eigvalsh(L + L.T - np.diag(np.diag(cov)))
    else:
        out.update(T=np.eye(4), cov=np.zeros((6, 6)), err=-1.0, cov_eig=np.zeros(6), status=status if status else 3)
    return out


def prm_dict(p):
    return dict(mode=p.mode, has_points=p.has_points, has_lines=p.has_lines, min_features=p.min_features,
                max_iters=p.max_iters, max_iters_ref=p.max_iters_ref, homog_th=p.homog_th, min_error=p.min_error,
                min_error_change=p.min_error_change, inlier_k=p.inlier_k)


def rot_angle(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1) / 2
    return float(np.arccos(min(1.0, max(-1.0, c))))
