"""Oracle-driven restatement of the per-frame loop (initialize / insertStereoPair / optimizePose /
updateFrame, /root/reference/src/stereoFrameHandler.cpp:35-180,307-392) — TEST INFRASTRUCTURE ONLY.
Every numeric step goes through oracle/stvo_oracle.c; this file is only the list/record plumbing."""
import numpy as np


def stereo_frame(orc, fr, cam, mp, has_points=True, has_lines=True):
    cols, rows = cam["width"], cam["height"]
    out = {}
    if has_points and len(fr["kp_l"]) and len(fr["kp_r"]):
        sp = orc.stereo_points(fr["kp_l"], fr["oct_l"], fr["desc_l"], fr["kp_r"], fr["desc_r"], cols, rows, cam, mp)
        out.update(pl=sp["pl"], P=sp["P"], sigma2p=sp["sigma2"], pdesc=np.ascontiguousarray(fr["desc_l"][sp["src_idx"]]),
                   m12_raw_p=sp["m12_raw"])
    else:
        out.update(pl=np.zeros((0, 2)), P=np.zeros((0, 3)), sigma2p=np.zeros(0), pdesc=np.zeros((0, 32), np.uint8),
                   m12_raw_p=-np.ones(len(fr["kp_l"]), np.int32))
    if has_lines and len(fr["kl_l"]) and len(fr["kl_r"]):
        sl = orc.stereo_lines(fr["kl_l"], fr["ang_l"], fr["oct_ll"], fr["ldesc_l"], fr["kl_r"], fr["ldesc_r"], cols, rows, cam, mp)
        out.update(spl=sl["spl"], epl=sl["epl"], sP=sl["sP"], eP=sl["eP"], le=sl["le"], sigma2l=sl["sigma2"],
                   llevel=fr["oct_ll"][sl["src_idx"]], ldesc=np.ascontiguousarray(fr["ldesc_l"][sl["src_idx"]]),
                   m12_raw_l=sl["m12_raw"])
    else:
        z3 = np.zeros((0, 3)); z2 = np.zeros((0, 2))
        out.update(spl=z2, epl=z2.copy(), sP=z3, eP=z3.copy(), le=z3.copy(), sigma2l=np.zeros(0), llevel=np.zeros(0, np.int32),
                   ldesc=np.zeros((0, 32), np.uint8), m12_raw_l=-np.ones(len(fr["kl_l"]), np.int32))
    return out


def matched_line_sigma2(sigma2, level, lsd_scale):
    """LineFeature::safeCopy re-applies the level scaling (src/stereoFeatures.cpp:117-135)."""
    out = np.empty(len(sigma2))
    for i, (s, lv) in enumerate(zip(sigma2, level)):
        for _ in range(int(lv)):
            s *= lsd_scale
        out[i] = 1.0 / (s * s)
    return out


def match_fanout(orc, ex, d1, d2, nnr, best_lr):
    """StVO::match with its two directions as two tasks (src/matching.cpp:69-74: matchNNR 12 || 21 through std::async when
    lrInParallel), then the mutual check of :80-86.  Same result as orc.match(d1, d2, nnr, best_lr)."""
    if not best_lr:
        return orc.match(d1, d2, nnr, 0)
    f12 = ex.submit(orc.match, d1, d2, nnr, 0)
    f21 = ex.submit(orc.match, d2, d1, nnr, 0)
    m12 = f12.result()[0].copy()
    m21 = f21.result()[0]
    has = m12 >= 0
    keep = has & (m21[np.where(has, m12, 0)] == np.arange(len(m12)))
    m12[~keep] = -1
    return m12, int(keep.sum())


def run_sequence(orc, frames, cam, mp, prm, fast=dict(adaptive=True, th0=20, mn=7, mx=30, inc=5, feat=50, err=0.5), keyframes=None, fanout=None, motion_model=False):
    """motion_model: Config::useMotionModel() — the initial DT of optimizePose is prev_frame->DT (the increment COMMITTED for the previous
    pair) unless !isGoodSolution(prev_frame->DT, prev_frame->DT_cov, prev_frame->err_norm) (src/stereoFrameHandler.cpp:317-324).  After
    initialize prev_frame->DT = I (:45) while DT_cov / err_norm are uninitialised memory in the reference: either verdict gives I.
    keyframes: None, or dict(min_entropy_ratio, max_kf_t_dist, max_kf_r_dist) to run needNewKF / currFrameIsKF after every
    optimizePose (src/stereoFrameHandler.cpp:1136-1218); every result then carries `new_kf`.
    fanout: None, or a concurrent.futures executor with >= 4 workers — the reference's own thread structure (points || lines in
    the stereo association and in f2fTracking, stereoFrame.cpp:64-72 / stereoFrameHandler.cpp:115-118, and 12 || 21 inside every
    match, matching.cpp:69-74): identical results, used by bench.py to time the CPU path at the reference's fan-out."""
    has_p, has_l = bool(prm.has_points), bool(prm.has_lines)
    kf_state = orc.kf_state() if keyframes else None
    def stereo(fr):
        if fanout is None or not (has_p and has_l):
            return stereo_frame(orc, fr, cam, mp, has_p, has_l)
        fp = fanout.submit(stereo_frame, orc, fr, cam, mp, True, False)   # points || lines
        fl = fanout.submit(stereo_frame, orc, fr, cam, mp, False, True)
        out = fp.result()
        ln = fl.result()
        out.update({k: ln[k] for k in ("spl", "epl", "sP", "eP", "le", "sigma2l", "llevel", "ldesc", "m12_raw_l")})
        return out

    def match(d1, d2, nnr):
        return orc.match(d1, d2, nnr, mp.best_lr_matches) if fanout is None else match_fanout(orc, fanout, d1, d2, nnr, mp.best_lr_matches)

    prev = stereo(frames[0])
    prev.update(Tfw=np.eye(4), Tfw_cov=np.eye(6), DT=np.eye(4), DT_cov=np.zeros((6, 6)), err_norm=-1.0)
    fast_th = fast["th0"]
    results = []
    for k in range(1, len(frames)):
        curr = stereo(frames[k])
        z3 = np.zeros((0, 3)); z2 = np.zeros((0, 2))
        rec = dict(P=z3, pl_obs=z2, sigma2p=np.zeros(0), inlier_p=np.zeros(0, np.int32), sP=z3, eP=z3, le_obs=z3, spl=z2,
                   epl=z2, sigma2l=np.zeros(0), inlier_l=np.zeros(0, np.int32))
        do_p = has_p and len(prev["P"]) and len(curr["P"])
        do_l = has_l and len(prev["sP"]) and len(curr["sP"])
        fut_l = None
        if fanout is not None and do_p and do_l:   # f2fTracking: the line matching runs beside the point matching
            fut_l = fanout.submit(match, prev["ldesc"], curr["ldesc"], mp.min_ratio_12_l)
        if do_p:
            m12, _ = match(prev["pdesc"], curr["pdesc"], mp.min_ratio_12_p)
            sel = np.nonzero(m12 >= 0)[0]
            rec.update(P=prev["P"][sel], pl_obs=curr["pl"][m12[sel]], sigma2p=prev["sigma2p"][sel], inlier_p=np.ones(len(sel), np.int32))
        if do_l:
            m12, _ = fut_l.result() if fut_l is not None else match(prev["ldesc"], curr["ldesc"], mp.min_ratio_12_l)
            sel = np.nonzero(m12 >= 0)[0]
            rec.update(sP=prev["sP"][sel], eP=prev["eP"][sel], le_obs=curr["le"][m12[sel]], spl=prev["spl"][sel],
                       epl=prev["epl"][sel], sigma2l=matched_line_sigma2(prev["sigma2l"][sel], prev["llevel"][sel], mp.lsd_scale),
                       inlier_l=np.ones(len(sel), np.int32))
        init_T = np.eye(4)
        if motion_model and orc.is_good(prev["DT"], prev["DT_cov"], prev["err_norm"]):   # :317-324
            init_T = prev["DT"]
        out = orc.optimize_pose(init_T, cam, prm, rec)
        curr.update(DT=np.array(out["T"], float).reshape(4, 4), DT_cov=np.array(out["cov"], float).reshape(6, 6), err_norm=float(out["err"]))
        if out["status"] == 0:
            curr["Tfw"] = orc.expmap(orc.logmap(prev["Tfw"] @ out["T"]))
            curr["Tfw_cov"] = orc.unccomp(prev["Tfw"], prev["Tfw_cov"], out["cov"])
        else:
            curr["Tfw"], curr["Tfw_cov"] = prev["Tfw"], prev["Tfw_cov"]
        n_inl_pt = out["n_inliers_pt"]
        if fast["adaptive"]:  # updateFrame, :66-86
            if np.array_equal(out["T"], np.eye(4)) or out["err"] > np.float32(fast["err"]):
                fast_th = max(fast["mn"], fast_th - 2 * fast["inc"])
            elif n_inl_pt < fast["feat"]:
                fast_th = max(fast["mn"], fast_th - 2 * fast["inc"])
            elif n_inl_pt < fast["feat"] * 2:
                fast_th = max(fast["mn"], fast_th - fast["inc"])
            elif n_inl_pt > fast["feat"] * 3:
                fast_th = min(fast["mx"], fast_th + fast["inc"])
        out.update(Tfw=curr["Tfw"], Tfw_cov=curr["Tfw_cov"], n_stereo_pt=len(curr["P"]), n_stereo_ls=len(curr["sP"]),
                   n_matched_pt=len(rec["sigma2p"]), n_matched_ls=len(rec["sigma2l"]), fast=fast_th)
        if keyframes:
            out["new_kf"] = orc.need_new_kf(kf_state, curr["Tfw"], out["T"], out["cov"], **keyframes)
            if out["new_kf"]:  # currFrameIsKF: the map frame restarts at this frame
                orc.curr_frame_is_kf(kf_state)
                curr["Tfw"], curr["Tfw_cov"] = np.eye(4), np.eye(6)
        results.append(out)
        prev = curr
    return results
