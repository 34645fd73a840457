"""Seeded synthetic stereo-VO inputs shaped like the BASELINE.json configs.

There is no KITTI / EuRoC data and no OpenCV in the image, so every config is synthetic: feature
records (key-points, key-lines, 256-bit descriptors) are generated directly, i.e. the hot path's
inputs as they leave the reference's ORB / LSD+LBD front-end (out of scope).  numpy only.
"""
from __future__ import annotations

import numpy as np

# config/dataset_params/kitti00-02.yaml:2-13 (values quoted in SURVEY.md §5)
KITTI_CAM = dict(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, b=0.537165719, width=1241, height=376)
# config/dataset_params/kitti03.yaml and kitti04-10.yaml
KITTI03_CAM = dict(fx=721.5377, fy=721.5377, cx=609.5593, cy=172.854, b=0.537150588, width=1242, height=375)
KITTI04_CAM = dict(fx=707.0912, fy=707.0912, cx=601.8873, cy=183.1104, b=0.537150653, width=1226, height=370)
# EuRoC: rectified intrinsics come out of cv::stereoRectify at run time in the reference
# (src/pinholeStereoCamera.cpp:82-96) and cannot be reproduced here; values chosen by this build.
EUROC_CAM = dict(fx=435.2, fy=435.2, cx=367.45, cy=252.2, b=0.110078, width=752, height=480)

SEED0 = 20250227


def frame_seed(seq: int, frame: int) -> int:
    return SEED0 + 100003 * seq + frame


def cam_vec(cam) -> np.ndarray:
    return np.array([cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["b"]], dtype=np.float64)


def expmap_so3(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def back_project(cam, u, v, disp):
    bd = cam["b"] / disp
    return np.stack([bd * (u - cam["cx"]), bd * (v - cam["cy"]), bd * cam["fx"]], axis=-1)


def project(cam, P):
    return np.stack([cam["cx"] + cam["fx"] * P[..., 0] / P[..., 2], cam["cy"] + cam["fy"] * P[..., 1] / P[..., 2]], axis=-1)


def random_desc(rng, n):
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def flip_bits(rng, desc, p):
    """Each of the 256 bits flips independently w.p. p  (count ~ Binomial(256, p))."""
    mask_bits = rng.random((desc.shape[0], 256)) < p
    mask = np.packbits(mask_bits, axis=1, bitorder="little")
    return desc ^ mask


def clustered_desc(rng, n, cluster_frac=0.6, cluster_size=8, spread_p=0.06):
    """Descriptors that do NOT discriminate like i.i.d. bits: `cluster_frac` of the rows come in groups of ~`cluster_size`
    near-duplicates (a common centre with each bit flipped w.p. `spread_p`: intra-cluster distance ~ 2 p (1 - p) 256),
    as repeated structure (windows, foliage, road markings) produces in real ORB / LBD rows; the rest are i.i.d.
    A clustered row's second-best distance is that of a cluster sibling, i.e. of the order of a true match's distance, so
    the ratio test and the mutual check of StVO::match (src/matching.cpp:53-58,80-86) are decided by close calls."""
    d = random_desc(rng, n)
    n_cl = int(round(cluster_frac * n))
    n_groups = max(1, n_cl // cluster_size)
    centres = random_desc(rng, n_groups)
    member = rng.integers(0, n_groups, n_cl)
    d[:n_cl] = flip_bits(rng, centres[member], spread_p)
    return np.ascontiguousarray(d[rng.permutation(n)])


def random_motion(rng, t_fwd=(0.5, 1.5), w_sigma=0.01, t_sigma=0.02):
    w = rng.normal(0.0, w_sigma, 3)
    t = np.array([rng.normal(0, t_sigma), rng.normal(0, t_sigma), -rng.uniform(*t_fwd)])
    T = np.eye(4)
    T[:3, :3] = expmap_so3(w)
    T[:3, 3] = t
    return T


def make_f2f_points(seed, n=2000, cam=KITTI_CAM, track_frac=0.75, outlier_frac=0.15, flip_p=0.08, noise_px=0.5,
                    depth=(4.0, 80.0), octave_probs=None, scale_factor=1.2, edge=19, motion=None, desc_model="iid",
                    cluster_kw=None):
    """BASELINE config 2: one prev/curr pair of stereo-point sets for brute-force f2f matching +
    optimizePose.  Returns a dict of numpy arrays (see keys below)."""
    rng = np.random.default_rng(seed)
    W, Hh = cam["width"], cam["height"]
    u = rng.uniform(edge, W - edge, n)
    v = rng.uniform(edge, Hh - edge, n)
    z = rng.uniform(depth[0], depth[1], n)
    disp = cam["b"] * cam["fx"] / z
    P = back_project(cam, u, v, disp)
    T = random_motion(rng) if motion is None else motion
    if octave_probs is None:
        level = np.zeros(n, dtype=np.int32)
    else:
        level = rng.choice(len(octave_probs), size=n, p=octave_probs).astype(np.int32)
    sigma2 = 1.0 / (scale_factor ** level.astype(np.float64)) ** 2

    # desc_model "iid": 256 independent bits per row; "clustered": groups of near-duplicate rows
    prev_desc = random_desc(rng, n) if desc_model == "iid" else clustered_desc(rng, n, **(cluster_kw or {}))
    n_tr = int(round(track_frac * n))
    tracked = rng.permutation(n)[:n_tr]
    Pc = P[tracked] @ T[:3, :3].T + T[:3, 3]
    obs = project(cam, Pc) + rng.normal(0.0, noise_px, (n_tr, 2))
    n_out = int(round(outlier_frac * n_tr))
    out_sel = rng.permutation(n_tr)[:n_out]
    obs[out_sel, 0] = rng.uniform(0, W, n_out)
    obs[out_sel, 1] = rng.uniform(0, Hh, n_out)
    curr_desc = np.empty((n, 32), dtype=np.uint8)
    curr_pl = np.empty((n, 2), dtype=np.float64)
    curr_desc[:n_tr] = flip_bits(rng, prev_desc[tracked], flip_p)
    curr_pl[:n_tr] = obs
    n_new = n - n_tr
    curr_desc[n_tr:] = random_desc(rng, n_new) if desc_model == "iid" else clustered_desc(rng, n_new, **(cluster_kw or {}))
    curr_pl[n_tr:, 0] = rng.uniform(edge, W - edge, n_new)
    curr_pl[n_tr:, 1] = rng.uniform(edge, Hh - edge, n_new)
    perm = rng.permutation(n)
    curr_desc = np.ascontiguousarray(curr_desc[perm])
    curr_pl = np.ascontiguousarray(curr_pl[perm])
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    true_m12 = -np.ones(n, dtype=np.int32)
    true_m12[tracked] = inv[:n_tr]
    is_gross = np.zeros(n, dtype=bool)
    is_gross[tracked[out_sel]] = True
    return dict(prev_desc=prev_desc, prev_P=np.ascontiguousarray(P), prev_pl=np.stack([u, v], 1), prev_disp=disp,
                prev_sigma2=sigma2, prev_level=level, curr_desc=curr_desc, curr_pl=curr_pl, T_true=T,
                true_m12=true_m12, is_gross=is_gross, cam=cam)


def make_line_set(rng, n, cam, depth=(4.0, 80.0), len_px=(30.0, 300.0), min_dy=5.0):
    """n random 3-D segments visible in the left image: pixel end points, end-point disparities."""
    W, Hh = cam["width"], cam["height"]
    sp = np.empty((n, 2)); ep = np.empty((n, 2))
    k = 0
    while k < n:
        m = 2 * (n - k) + 8
        s = np.stack([rng.uniform(10, W - 10, m), rng.uniform(10, Hh - 10, m)], 1)
        L = rng.uniform(len_px[0], min(len_px[1], Hh * 0.9), m)
        a = rng.uniform(0, np.pi, m)
        e = s + np.stack([L * np.cos(a), L * np.sin(a)], 1)
        ok = (e[:, 0] > 10) & (e[:, 0] < W - 10) & (e[:, 1] > 10) & (e[:, 1] < Hh - 10) & (np.abs(e[:, 1] - s[:, 1]) > min_dy)
        s, e = s[ok], e[ok]
        take = min(n - k, len(s))
        sp[k:k + take] = s[:take]; ep[k:k + take] = e[:take]
        k += take
    zs = rng.uniform(depth[0], depth[1], n)
    ze = zs * rng.uniform(0.75, 1.0 / 0.75, n)  # end-point depth ratio >= 0.75
    sdisp = cam["b"] * cam["fx"] / zs
    edisp = cam["b"] * cam["fx"] / ze
    return sp, ep, sdisp, edisp


def line_eq(sp, ep):
    """Normalised homogeneous line through two pixel points (src/stereoFrame.cpp:353-355)."""
    s = np.concatenate([sp, np.ones((len(sp), 1))], 1)
    e = np.concatenate([ep, np.ones((len(ep), 1))], 1)
    le = np.cross(s, e)
    return le / np.sqrt(le[:, 0:1] ** 2 + le[:, 1:2] ** 2)


def make_matched_records(seed, n_pts=1200, n_lines=60, cam=KITTI_CAM, outlier_frac=0.15, noise_px=0.5,
                         depth=(4.0, 80.0), octave_probs=None, scale_factor=1.2, motion=None):
    """Already-associated correspondence records (matched_pt / matched_ls) for optimizer-only tests:
    what f2fTracking hands to optimizePose (src/stereoFrameHandler.cpp:144-152,167-179)."""
    rng = np.random.default_rng(seed)
    W, Hh = cam["width"], cam["height"]
    T = random_motion(rng) if motion is None else motion
    u = rng.uniform(19, W - 19, n_pts); v = rng.uniform(19, Hh - 19, n_pts)
    z = rng.uniform(depth[0], depth[1], n_pts)
    P = back_project(cam, u, v, cam["b"] * cam["fx"] / z)
    obs = project(cam, P @ T[:3, :3].T + T[:3, 3]) + rng.normal(0, noise_px, (n_pts, 2))
    n_out = int(round(outlier_frac * n_pts))
    sel = rng.permutation(n_pts)[:n_out]
    obs[sel, 0] = rng.uniform(0, W, n_out); obs[sel, 1] = rng.uniform(0, Hh, n_out)
    if octave_probs is None:
        lvl = np.zeros(n_pts, dtype=np.int64)
    else:
        lvl = rng.choice(len(octave_probs), size=n_pts, p=octave_probs)
    sigma2p = 1.0 / (scale_factor ** lvl.astype(np.float64)) ** 2
    rec = dict(P=np.ascontiguousarray(P), pl_obs=np.ascontiguousarray(obs), sigma2p=sigma2p,
               inlier_p=np.ones(n_pts, dtype=np.int32), T_true=T, cam=cam)
    if n_lines > 0:
        sp, ep, sd, ed = make_line_set(rng, n_lines, cam, depth)
        sP = back_project(cam, sp[:, 0], sp[:, 1], sd)
        eP = back_project(cam, ep[:, 0], ep[:, 1], ed)
        sp_c = project(cam, sP @ T[:3, :3].T + T[:3, 3]) + rng.normal(0, noise_px, (n_lines, 2))
        ep_c = project(cam, eP @ T[:3, :3].T + T[:3, 3]) + rng.normal(0, noise_px, (n_lines, 2))
        n_lo = int(round(outlier_frac * n_lines))
        lsel = rng.permutation(n_lines)[:n_lo]
        sp_c[lsel] += rng.uniform(-60, 60, (n_lo, 2)); ep_c[lsel] += rng.uniform(-60, 60, (n_lo, 2))
        rec.update(sP=np.ascontiguousarray(sP), eP=np.ascontiguousarray(eP), le_obs=np.ascontiguousarray(line_eq(sp_c, ep_c)),
                   spl=np.ascontiguousarray(sp), epl=np.ascontiguousarray(ep), sigma2l=np.ones(n_lines),
                   inlier_l=np.ones(n_lines, dtype=np.int32))
    else:
        z3 = np.zeros((0, 3)); z2 = np.zeros((0, 2))
        rec.update(sP=z3, eP=z3.copy(), le_obs=z3.copy(), spl=z2, epl=z2.copy(), sigma2l=np.zeros(0),
                   inlier_l=np.zeros(0, dtype=np.int32))
    return rec


# ------------------------------------------------------------------------------------------------
# Stereo feature SEQUENCES (what the out-of-scope ORB / LSD+LBD front-end would hand to the hot path)
# ------------------------------------------------------------------------------------------------
def make_stereo_sequence(seed, n_frames=6, n_pts=600, n_lines=60, cam=KITTI_CAM, distract=0.2, flip_p=0.03,
                         noise_px=0.3, depth=(4.0, 60.0), octave_probs=None, outlier_frac=0.05, cluster_kw=None):
    """A persistent 3-D landmark world observed by a forward-moving rectified stereo rig.
    Returns a list of per-frame dicts: kp_l/kp_r (float32 [n,2]), oct_l, desc_l/desc_r (uint8 [n,32]),
    kl_l/kl_r (float32 [n,4] sx,sy,ex,ey), ang_l, oct_ll, ldesc_l/ldesc_r, plus T_true (prev->curr).
    cluster_kw (None: i.i.d. descriptor bits): dict(cluster_frac, cluster_size, spread_p) — the landmark descriptors (points and
    lines) come in groups of near-duplicates around a fixed set of centres, as clustered_desc describes: repeated structure, so that
    ratio tests and the mutual check are decided by close calls in BOTH matchers (stereo grid and f2f)."""
    rng = np.random.default_rng(seed)
    W, Hh = cam["width"], cam["height"]
    edge = 19.0
    if cluster_kw is None:
        land_desc_p = land_desc_l = lambda n: random_desc(rng, n)
    else:
        frac, size, sp = cluster_kw["cluster_frac"], cluster_kw["cluster_size"], cluster_kw["spread_p"]

        def make_model(n_total):
            centres = random_desc(rng, max(1, int(round(frac * n_total)) // size))

            def draw(n):
                d = random_desc(rng, n)
                if n:
                    cl = rng.random(n) < frac
                    k = int(cl.sum())
                    if k:
                        d[cl] = flip_bits(rng, centres[rng.integers(0, len(centres), k)], sp)
                return d
            return draw
        land_desc_p, land_desc_l = make_model(n_pts), make_model(max(n_lines, 1))

    def new_points(n, Tcw):
        u = rng.uniform(edge, W - edge, n); v = rng.uniform(edge, Hh - edge, n); z = rng.uniform(depth[0], depth[1], n)
        Pc = back_project(cam, u, v, cam["b"] * cam["fx"] / z)
        Tinv = np.linalg.inv(Tcw)
        return Pc @ Tinv[:3, :3].T + Tinv[:3, 3]

    def new_lines(n, Tcw):
        sp, ep, sd, ed = make_line_set(rng, n, cam, depth)
        Tinv = np.linalg.inv(Tcw)
        s = back_project(cam, sp[:, 0], sp[:, 1], sd) @ Tinv[:3, :3].T + Tinv[:3, 3]
        e = back_project(cam, ep[:, 0], ep[:, 1], ed) @ Tinv[:3, :3].T + Tinv[:3, 3]
        return s, e

    Tcw = np.eye(4)
    Pw = new_points(n_pts, Tcw)
    pdesc = land_desc_p(n_pts)
    plevel = (np.zeros(n_pts, np.int32) if octave_probs is None
              else rng.choice(len(octave_probs), size=n_pts, p=octave_probs).astype(np.int32))
    Ls, Le = new_lines(n_lines, Tcw) if n_lines else (np.zeros((0, 3)), np.zeros((0, 3)))
    ldesc = land_desc_l(n_lines)
    frames = []
    for k in range(n_frames):
        T_step = np.eye(4)
        if k > 0:
            T_step = random_motion(rng)
            Tcw = T_step @ Tcw
        # replenish landmarks that left the field of view
        Pc = Pw @ Tcw[:3, :3].T + Tcw[:3, 3]
        uv = project(cam, Pc)
        vis = (Pc[:, 2] > 2.5) & (uv[:, 0] > edge) & (uv[:, 0] < W - edge) & (uv[:, 1] > edge) & (uv[:, 1] < Hh - edge)
        nb = int((~vis).sum())
        if nb:
            Pw[~vis] = new_points(nb, Tcw)
            pdesc[~vis] = land_desc_p(nb)
            Pc = Pw @ Tcw[:3, :3].T + Tcw[:3, 3]
            uv = project(cam, Pc)
        if n_lines:
            sc = Ls @ Tcw[:3, :3].T + Tcw[:3, 3]; ec = Le @ Tcw[:3, :3].T + Tcw[:3, 3]
            su = project(cam, sc); eu = project(cam, ec)
            lvis = (sc[:, 2] > 2.5) & (ec[:, 2] > 2.5)
            for arr in (su, eu):
                lvis &= (arr[:, 0] > 5) & (arr[:, 0] < W - 5) & (arr[:, 1] > 5) & (arr[:, 1] < Hh - 5)
            lvis &= np.abs(su[:, 1] - eu[:, 1]) > 3.0
            nb = int((~lvis).sum())
            if nb:
                s_new, e_new = new_lines(nb, Tcw)
                Ls[~lvis] = s_new; Le[~lvis] = e_new
                ldesc[~lvis] = land_desc_l(nb)
                sc = Ls @ Tcw[:3, :3].T + Tcw[:3, 3]; ec = Le @ Tcw[:3, :3].T + Tcw[:3, 3]
                su = project(cam, sc); eu = project(cam, ec)
        # ---- points: left observation, right = same row shifted by the disparity
        n_d = int(round(distract * n_pts))
        obs = uv + rng.normal(0, noise_px, uv.shape)
        n_out = int(round(outlier_frac * n_pts))  # gross mis-detections (wrong geometry, right descriptor)
        sel = rng.permutation(n_pts)[:n_out]
        obs[sel, 0] = rng.uniform(edge, W - edge, n_out); obs[sel, 1] = rng.uniform(edge, Hh - edge, n_out)
        disp = cam["b"] * cam["fx"] / Pc[:, 2] + rng.normal(0, 0.05, n_pts)
        kp_l = np.concatenate([obs, np.stack([rng.uniform(edge, W - edge, n_d), rng.uniform(edge, Hh - edge, n_d)], 1)]).astype(np.float32)
        kp_r = np.empty((n_pts + n_d, 2), np.float32)
        kp_r[:n_pts, 0] = (kp_l[:n_pts, 0].astype(np.float64) - disp).astype(np.float32)
        kp_r[:n_pts, 1] = kp_l[:n_pts, 1]  # rectified: identical row (config_kitti max_dist_epip = 0)
        kp_r[n_pts:, 0] = rng.uniform(edge, W - edge, n_d); kp_r[n_pts:, 1] = rng.uniform(edge, Hh - edge, n_d)
        desc_l = np.concatenate([flip_bits(rng, pdesc, flip_p), random_desc(rng, n_d)])
        desc_r = np.concatenate([flip_bits(rng, pdesc, flip_p), random_desc(rng, n_d)])
        oct_l = np.concatenate([plevel, np.zeros(n_d, np.int32)])
        pl_ = rng.permutation(n_pts + n_d); pr_ = rng.permutation(n_pts + n_d)
        fr = dict(kp_l=np.ascontiguousarray(kp_l[pl_]), oct_l=np.ascontiguousarray(oct_l[pl_]), desc_l=np.ascontiguousarray(desc_l[pl_]),
                  kp_r=np.ascontiguousarray(kp_r[pr_]), desc_r=np.ascontiguousarray(desc_r[pr_]), T_true=T_step, Tcw=Tcw.copy())
        # ---- lines
        if n_lines:
            n_dl = int(round(distract * n_lines))
            so = su + rng.normal(0, noise_px, su.shape); eo = eu + rng.normal(0, noise_px, eu.shape)
            sd = cam["b"] * cam["fx"] / sc[:, 2]; ed = cam["b"] * cam["fx"] / ec[:, 2]
            dsp, dep, _, _ = make_line_set(rng, max(n_dl, 1), cam, depth)
            dsp, dep = dsp[:n_dl], dep[:n_dl]
            kl_l = np.concatenate([np.concatenate([so, eo], 1), np.concatenate([dsp, dep], 1)]).astype(np.float32)
            kl_r = kl_l.copy()
            kl_r[:n_lines, 0] = (kl_l[:n_lines, 0].astype(np.float64) - sd).astype(np.float32)
            kl_r[:n_lines, 2] = (kl_l[:n_lines, 2].astype(np.float64) - ed).astype(np.float32)
            kl_r[n_lines:] += rng.uniform(-80, 80, (n_dl, 4)).astype(np.float32)
            ldesc_l = np.concatenate([flip_bits(rng, ldesc, flip_p), random_desc(rng, n_dl)])
            ldesc_r = np.concatenate([flip_bits(rng, ldesc, flip_p), random_desc(rng, n_dl)])
            ang = np.arctan2(kl_l[:, 3] - kl_l[:, 1], kl_l[:, 2] - kl_l[:, 0]).astype(np.float32)
            ll_ = rng.permutation(n_lines + n_dl); lr_ = rng.permutation(n_lines + n_dl)
            fr.update(kl_l=np.ascontiguousarray(kl_l[ll_]), ang_l=np.ascontiguousarray(ang[ll_]),
                      oct_ll=np.zeros(n_lines + n_dl, np.int32), ldesc_l=np.ascontiguousarray(ldesc_l[ll_]),
                      kl_r=np.ascontiguousarray(kl_r[lr_]), ldesc_r=np.ascontiguousarray(ldesc_r[lr_]))
        else:
            z4 = np.zeros((0, 4), np.float32); zd = np.zeros((0, 32), np.uint8)
            fr.update(kl_l=z4, ang_l=np.zeros(0, np.float32), oct_ll=np.zeros(0, np.int32), ldesc_l=zd, kl_r=z4.copy(), ldesc_r=zd.copy())
        frames.append(fr)
    return frames


def write_sequence(path, frames, cam):
    """Binary hand-over file read by stvo-pl_amd/app/imagesStVO_synth.cpp."""
    import struct
    with open(path, "wb") as f:
        f.write(b"STVOSEQ1")
        f.write(struct.pack("<iii", len(frames), cam["width"], cam["height"]))
        f.write(struct.pack("<5d", cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["b"]))
        for fr in frames:
            f.write(struct.pack("<4i", len(fr["kp_l"]), len(fr["kp_r"]), len(fr["kl_l"]), len(fr["kl_r"])))
            for kp, octv, desc in ((fr["kp_l"], fr["oct_l"], fr["desc_l"]), (fr["kp_r"], np.zeros(len(fr["kp_r"]), np.int32), fr["desc_r"])):
                rec = np.zeros(len(kp), dtype=[("x", "<f4"), ("y", "<f4"), ("o", "<i4")])
                rec["x"], rec["y"], rec["o"] = kp[:, 0], kp[:, 1], octv
                f.write(rec.tobytes()); f.write(np.ascontiguousarray(desc, np.uint8).tobytes())
            for kl, ang, octv, desc in ((fr["kl_l"], fr["ang_l"], fr["oct_ll"], fr["ldesc_l"]),
                                        (fr["kl_r"], np.zeros(len(fr["kl_r"]), np.float32), np.zeros(len(fr["kl_r"]), np.int32), fr["ldesc_r"])):
                rec = np.zeros(len(kl), dtype=[("sx", "<f4"), ("sy", "<f4"), ("ex", "<f4"), ("ey", "<f4"), ("a", "<f4"), ("o", "<i4")])
                if len(kl):
                    rec["sx"], rec["sy"], rec["ex"], rec["ey"], rec["a"], rec["o"] = kl[:, 0], kl[:, 1], kl[:, 2], kl[:, 3], ang, octv
                f.write(rec.tobytes()); f.write(np.ascontiguousarray(desc, np.uint8).tobytes())


def write_image_sequence(path, pairs, cam):
    """Stereo IMAGE hand-over file read by stvo-pl_amd/app/imagesStVO_synth.cpp ("STVOIMG1"): pairs = [(left, right), ...] uint8."""
    import struct
    with open(path, "wb") as f:
        f.write(b"STVOIMG1")
        f.write(struct.pack("<iii", len(pairs), cam["width"], cam["height"]))
        f.write(struct.pack("<5d", cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["b"]))
        for left, right in pairs:
            assert left.shape == (cam["height"], cam["width"]) and right.shape == left.shape
            f.write(np.ascontiguousarray(left, np.uint8).tobytes()); f.write(np.ascontiguousarray(right, np.uint8).tobytes())


RESULT_DTYPE = np.dtype([("ints", "<i4", (12,)), ("DT", "<f8", (16,)), ("DT_cov", "<f8", (36,)), ("cov_eig", "<f8", (6,)),
                         ("err", "<f8"), ("Tfw", "<f8", (16,)), ("Tfw_cov", "<f8", (36,)), ("fast", "<i4"), ("pad", "<i4")])


def read_results(path):
    return np.fromfile(path, dtype=RESULT_DTYPE)


def make_f2f_points_lines(seed, n=2000, n_lines=100, cam=KITTI_CAM, track_frac=0.75, **kw):
    """BASELINE config 3/4 shape for the f2f + optimizePose stage: the point sets of make_f2f_points plus
    prev/curr stereo-line sets (prev: 3-D end points, left-image end points, LBD rows; curr: line equations
    of the re-observed segments, LBD rows, permuted, padded with unrelated lines)."""
    fr = make_f2f_points(seed, n=n, cam=cam, track_frac=track_frac, **kw)
    rng = np.random.default_rng(seed + 7919)
    T = fr["T_true"]
    sp, ep, sd, ed = make_line_set(rng, n_lines, cam)
    sP = back_project(cam, sp[:, 0], sp[:, 1], sd); eP = back_project(cam, ep[:, 0], ep[:, 1], ed)
    prev_ldesc = random_desc(rng, n_lines)
    n_tr = int(round(track_frac * n_lines))
    tracked = rng.permutation(n_lines)[:n_tr]
    so = project(cam, sP[tracked] @ T[:3, :3].T + T[:3, 3]) + rng.normal(0, 0.5, (n_tr, 2))
    eo = project(cam, eP[tracked] @ T[:3, :3].T + T[:3, 3]) + rng.normal(0, 0.5, (n_tr, 2))
    n_out = int(round(0.15 * n_tr))
    sel = rng.permutation(n_tr)[:n_out]
    so[sel] += rng.uniform(-60, 60, (n_out, 2)); eo[sel] += rng.uniform(-60, 60, (n_out, 2))
    n_new = n_lines - n_tr
    nsp, nep, _, _ = make_line_set(rng, max(n_new, 1), cam)
    curr_le = np.concatenate([line_eq(so, eo), line_eq(nsp[:n_new], nep[:n_new])])
    curr_ldesc = np.concatenate([flip_bits(rng, prev_ldesc[tracked], 0.08), random_desc(rng, n_new)])
    perm = rng.permutation(n_lines)
    fr.update(prev_ldesc=prev_ldesc, prev_sP=np.ascontiguousarray(sP), prev_eP=np.ascontiguousarray(eP),
              prev_spl=np.ascontiguousarray(sp), prev_epl=np.ascontiguousarray(ep), prev_sigma2l=np.ones(n_lines),
              curr_le=np.ascontiguousarray(curr_le[perm]), curr_ldesc=np.ascontiguousarray(curr_ldesc[perm]))
    return fr


# ------------------------------------------------------------------------------------------------
# BASELINE configs[4] / SURVEY.md §8(d) config 5: KITTI sequences 00-07 side by side, one calibration per dataset
# (config/dataset_params/kitti00-02.yaml, kitti03.yaml, kitti04-10.yaml)
# ------------------------------------------------------------------------------------------------
CONFIG5_N_SEQUENCES = 8
CONFIG5_CAMS = [KITTI_CAM] * 3 + [KITTI03_CAM] + [KITTI04_CAM] * 4


def config5_cam(seq: int):
    return CONFIG5_CAMS[seq % CONFIG5_N_SEQUENCES]


def make_config5_sequence(seq: int, n_frames: int, n_pts=1650, n_lines=85, replica=0, **kw):
    """Sequence `seq` (0..7) of config 5: generated as config 3 (points + lines, make_stereo_sequence) with the camera of
    that KITTI sequence.  `replica` > 0 gives further independent streams of the same shape (bench: B streams per GPU)."""
    return make_stereo_sequence(frame_seed(seq + CONFIG5_N_SEQUENCES * replica, 0), n_frames=n_frames, n_pts=n_pts,
                                n_lines=n_lines, cam=config5_cam(seq), **kw)


# ------------------------------------------------------------------------------------------------
# Synthetic grey images for the ORB front-end (there is no KITTI / EuRoC data in the image): overlapping rectangles and
# discs of random brightness (corners and edges for FAST), a smooth illumination gradient and sensor noise
# ------------------------------------------------------------------------------------------------
def make_image(seed, cols=1241, rows=376, n_rects=1000, n_discs=250, noise=3.0):
    rng = np.random.default_rng(seed)
    img = np.full((rows, cols), 110.0)
    yy, xx = np.mgrid[0:rows, 0:cols]
    img += 25.0 * np.sin(xx / cols * 3.1) + 18.0 * np.cos(yy / rows * 2.3)
    for _ in range(n_rects):
        w, h = rng.integers(6, 90), rng.integers(6, 70)
        x0, y0 = rng.integers(-20, cols), rng.integers(-20, rows)
        img[max(y0, 0):max(y0 + h, 0), max(x0, 0):max(x0 + w, 0)] = rng.uniform(15, 240)
    for _ in range(n_discs):
        cx, cy, r = rng.uniform(0, cols), rng.uniform(0, rows), rng.uniform(3, 22)
        m = (xx - cx) ** 2 + (yy - cy) ** 2 <= r * r
        img[m] = rng.uniform(15, 240)
    img += rng.normal(0.0, noise, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_stereo_image_sequence(seed, n_frames, cam, disparities=(8, 12, 16, 24, 32), shift_per_disp=0.3, cover=0.3):
    """Rectified stereo IMAGE pairs of a camera translating along +x past fronto-parallel textured layers: a layer of disparity d
    appears d pixels further left in the right image and moves round(shift_per_disp * d) pixels to the left per frame (integer
    shifts, so every view is an exact crop — no resampling; the rounding leaves residuals of a few tenths of a pixel, like the
    quantisation of real key-points).  Nearer layers cover `cover` of the layer behind them.  Ground truth: tx ~ shift_per_disp * b
    per frame.  Returns [(left, right)] uint8 [rows, cols]."""
    cols, rows = cam["width"], cam["height"]
    disparities = sorted(disparities)
    shifts = [int(round(shift_per_disp * d)) for d in disparities]
    margin = max(d + n_frames * sh for d, sh in zip(disparities, shifts)) + 8
    W = cols + margin
    rng = np.random.default_rng(seed + 1)
    layers, masks = [], []
    for li, d in enumerate(disparities):
        layers.append(make_image(seed + 7919 * li, W, rows, n_rects=int(1000 * W / 1241), n_discs=int(250 * W / 1241)))
        m = np.zeros((rows, W), bool)
        if li > 0:  # blobs of this layer in front of everything farther away
            while m.mean() < cover:
                w, h = rng.integers(40, 160), rng.integers(30, 100)
                x0, y0 = rng.integers(0, W - w), rng.integers(0, rows - h)
                m[y0:y0 + h, x0:x0 + w] = True
        else:
            m[:] = True
        masks.append(m)
    out = []
    for k in range(n_frames):
        views = []
        for right in (0, 1):
            v = None
            for layer, m, d, sh in zip(layers, masks, disparities, shifts):
                o = k * sh + right * d
                if v is None:
                    v = layer[:, o:o + cols].copy()
                else:
                    mm = m[:, o:o + cols]
                    v[mm] = layer[:, o:o + cols][mm]
            views.append(v)
        out.append((views[0], views[1]))
    return out
