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


def random_motion(rng, t_fwd=(0.5, 1.5), w_sigma=0.01, t_sigma=0.02):
    w = rng.normal(0.0, w_sigma, 3)
    t = np.array([rng.normal(0, t_sigma), rng.normal(0, t_sigma), -rng.uniform(*t_fwd)])
    T = np.eye(4)
    T[:3, :3] = expmap_so3(w)
    T[:3, 3] = t
    return T


def make_f2f_points(seed, n=2000, cam=KITTI_CAM, track_frac=0.75, outlier_frac=0.15, flip_p=0.08, noise_px=0.5,
                    depth=(4.0, 80.0), octave_probs=None, scale_factor=1.2, edge=19, motion=None):
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

    prev_desc = random_desc(rng, n)
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
    curr_desc[n_tr:] = random_desc(rng, n_new)
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
