"""GPU parity of the device-resident per-frame pipeline (stvo_seq_*: stereo association on the grid, tail filters,
record building, f2f tracking and optimizePose all on the device, state kept in HBM across frames) against the
oracle-driven pipeline, for several independent sequences advancing in lock-step."""
import numpy as np
import pytest

import np_model
import pipeline_ref
from stvo_amd import synth
from stvo_amd.ctypes_types import match_params, opt_params

pytestmark = pytest.mark.gpu


def run_and_compare(oracle, seqs, cam, preset, mode=0, has_lines=1, max_kp=2048, max_kl=320, motion_model=False):
    from stvo_amd import capi
    B, nf = len(seqs), len(seqs[0])
    mp = match_params(preset)
    op = opt_params(preset, mode=mode, has_lines=has_lines)
    cams = [cam] * B if isinstance(cam, dict) else list(cam)   # one calibration for all, or one per sequence
    ctx = capi.Context(device_id=0, max_rows=2048, max_batch=max(B, 1))
    dev = capi.Sequences(ctx, B, max_kp, max_kl, cam, mp, op)
    try:
        if motion_model:
            dev.set_motion_model(True)
        refs = [pipeline_ref.run_sequence(oracle, seqs[b], cams[b], mp, op, motion_model=motion_model) for b in range(B)]
        ref0 = [pipeline_ref.stereo_frame(oracle, seqs[b][0], cams[b], mp, True, bool(has_lines)) for b in range(B)]
        for k in range(nf):
            res, counts = dev.push([seqs[b][k] for b in range(B)])
            for b in range(B):
                if k == 0:
                    assert counts[b, 0] == len(ref0[b]["P"]) and counts[b, 1] == len(ref0[b]["sP"])
                    continue
                o = refs[b][k - 1]
                r = res[b]
                assert counts[b, 0] == o["n_stereo_pt"] and counts[b, 1] == o["n_stereo_ls"], (b, k, counts[b], o["n_stereo_pt"], o["n_stereo_ls"])
                assert r["n_matched_pt"] == o["n_matched_pt"] and r["n_matched_ls"] == o["n_matched_ls"]
                assert r["status"] == o["status"] and r["path"] == o["path"] and tuple(r["iters"]) == o["iters"], (b, k)
                assert r["n_inliers_pt"] == o["n_inliers_pt"] and r["n_inliers_ls"] == o["n_inliers_ls"]
                T = r["T"].reshape(4, 4)
                assert np_model.rot_angle(T[:3, :3], o["T"][:3, :3]) < 1e-4 and np.linalg.norm(T[:3, 3] - o["T"][:3, 3]) < 1e-3
                assert np.allclose(T, o["T"], atol=1e-8) and np.isclose(r["err"], o["err"], rtol=1e-8)
                assert np.allclose(r["cov"].reshape(6, 6), o["cov"], rtol=1e-6, atol=1e-12)
    finally:
        dev.close()
        ctx.close()


@pytest.mark.parametrize("env", [{}, {"STVO_POSE_LOS": "0"}], ids=["lines-on-solver-wave", "lines-on-worker-waves"])
def test_seq_pipeline_kitti_points_and_lines(oracle, switches, env):
    """Few frame pairs: the latency pose kernel.  Its key-lines (60-80 here) are evaluated by the solver wave by default, by the worker
    waves with STVO_POSE_LOS=0 (and whenever a pair has more than 128): the same results either way."""
    switches(env)
    cam = synth.KITTI_CAM
    seqs = [synth.make_stereo_sequence(500 + b, n_frames=5, n_pts=600 + 150 * b, n_lines=60 + 10 * b, cam=cam) for b in range(3)]
    run_and_compare(oracle, seqs, cam, "kitti")


def test_seq_pipeline_kitti_points_only_2000(oracle):
    cam = synth.KITTI_CAM
    seqs = [synth.make_stereo_sequence(600 + b, n_frames=4, n_pts=1650, n_lines=0, cam=cam) for b in range(2)]
    run_and_compare(oracle, seqs, cam, "kitti", has_lines=0)


@pytest.mark.parametrize("B,pose", [(3, None), (40, "4"), (40, "1")], ids=["latency-kernel", "batch-kernel", "latency-kernel-40"])
def test_seq_pipeline_motion_model(oracle, switches, B, pose):
    """use_motion_model = true (stvo_seq_set_motion_model): the initial DT of a pair is the increment committed for the previous pair
    — the rule of src/stereoFrameHandler.cpp:317-324 applied on the device by the previous step's commit — against the oracle
    pipeline with the same rule, for both pose kernels.  The committed increment is the INVERSE of the optimiser's variable (:374), so
    under forward motion the reference starts every pair about two increments away from the answer: iteration counts and paths differ
    from the identity start, and the one sequence whose second frame carries no features exercises 'previous pair rejected -> I'."""
    if pose:
        switches({"STVO_POSE_KERNEL": pose})
    cam = synth.KITTI_CAM
    seqs = [synth.make_stereo_sequence(900 + b, n_frames=5, n_pts=300 + 40 * (b % 7), n_lines=30 + 5 * (b % 4), cam=cam) for b in range(B)]
    z2 = np.zeros((0, 2), np.float32); zd = np.zeros((0, 32), np.uint8); z4 = np.zeros((0, 4), np.float32)
    seqs[1][2] = dict(seqs[1][2], kp_r=z2, desc_r=zd, kl_r=z4, ldesc_r=zd)   # right camera dropped out: pairs 2 and 3 are rejected
    run_and_compare(oracle, seqs, cam, "kitti", max_kp=1024, max_kl=128, motion_model=True)
    # the motion model really changes the computation: some pair takes a different number of evaluations than from the identity
    mp, op = match_params("kitti"), opt_params("kitti")
    a = pipeline_ref.run_sequence(oracle, seqs[0], cam, mp, op, motion_model=True)
    b = pipeline_ref.run_sequence(oracle, seqs[0], cam, mp, op, motion_model=False)
    assert any(x["iters"] != y["iters"] for x, y in zip(a, b))


@pytest.mark.parametrize("B", [2, 24])
def test_seq_pipeline_clustered_descriptors(oracle, B):
    """The streams behind bench.py's value_clustered: landmark descriptors in groups of near-duplicates (60 % of the rows, ~8 per group,
    6 % of the bits spread), so that the ratio tests of BOTH matchers and the mutual check are decided by close calls and the reverse
    check runs its light and heavy scans (hamming_knn2_mfma_reverse_kernel) — the whole pipeline against the oracle, frame by frame,
    on the latency kernels (B = 2) and on the batch machinery (B = 24: K1m forward scan + plan + reverse scans + batch pose kernel)."""
    kw = dict(cluster_frac=0.6, cluster_size=8, spread_p=0.06)
    seqs = [synth.make_config5_sequence(b % 8, n_frames=4, n_pts=900 + 90 * (b % 5), n_lines=60, cluster_kw=kw) for b in range(B)]
    cams = [synth.config5_cam(b % 8) for b in range(B)]
    run_and_compare(oracle, seqs, cams, "kitti", max_kp=2048, max_kl=128)


@pytest.mark.parametrize("mode", [0, 2])
def test_seq_pipeline_euroc_line_heavy(oracle, mode):
    cam = synth.EUROC_CAM
    seqs = [synth.make_stereo_sequence(700 + b + mode, n_frames=4, n_pts=500, n_lines=200, cam=cam, depth=(1.0, 8.0),
                                       octave_probs=[.5, .25, .15, .1], outlier_frac=0.2) for b in range(2)]
    run_and_compare(oracle, seqs, cam, "euroc", mode=mode)


@pytest.mark.parametrize("nw", ["2", "4"])
def test_seq_pipeline_compact_pose_kernel_full_frames_and_high_levels(oracle, switches, nw):
    """The batch pose kernel on compact records (pose2c_kernel, forced here for a small batch) away from its comfortable shape:
    frames with ~2000 stereo points (a thread owns up to 16 prev points: the 12 LDS planes AND the register-resident record AND
    the off-chip path are all in use, and the inliers do not fit the planes' dense re-deal), and pyramid levels beyond the 15
    the sigma table holds (those records are gathered at every use).  Everything against the oracle-driven loop."""
    switches({"STVO_POSE_KERNEL": "4", "STVO_POSE2P_NW": nw})
    cam = synth.KITTI_CAM
    full = [synth.make_stereo_sequence(1900 + b, n_frames=4, n_pts=1690, n_lines=40, cam=cam) for b in range(2)]
    run_and_compare(oracle, full, cam, "kitti")
    probs = [0.4, 0.2, 0.1, 0.05] + [0.0] * 10 + [0.1, 0.1, 0.05]   # levels 14, 15, 16 occur
    deep = [synth.make_stereo_sequence(1950 + b, n_frames=4, n_pts=700, n_lines=30, cam=cam, octave_probs=probs) for b in range(2)]
    run_and_compare(oracle, deep, cam, "kitti")


def test_seq_pipeline_empty_and_tiny_frames(oracle):
    """Frames with no right features, no lines, or too few features: in-band failures, no crashes."""
    cam = synth.KITTI_CAM
    seq = synth.make_stereo_sequence(800, n_frames=4, n_pts=300, n_lines=30, cam=cam)
    z2 = np.zeros((0, 2), np.float32); zd = np.zeros((0, 32), np.uint8); z4 = np.zeros((0, 4), np.float32)
    seq[2] = dict(seq[2], kp_r=z2, desc_r=zd, kl_r=z4, ldesc_r=zd)   # right camera dropped out
    tiny = synth.make_stereo_sequence(801, n_frames=4, n_pts=6, n_lines=0, cam=cam, distract=0.0)
    run_and_compare(oracle, [seq, tiny], cam, "kitti")


def test_seq_pipeline_line_stage_skipped_and_resumed(oracle):
    """A frame without key-lines skips the line stage (and the f2f line matching on both sides of it); the stage
    must pick up again on the next frame with lines.  Single sequence, so the zero-copy read-back path is used."""
    cam = synth.KITTI_CAM
    seq = synth.make_stereo_sequence(900, n_frames=6, n_pts=500, n_lines=50, cam=cam)
    zd = np.zeros((0, 32), np.uint8); z4 = np.zeros((0, 4), np.float32); zi = np.zeros(0, np.int32)
    for k in (1, 4):  # the line detector found nothing in these frames
        seq[k] = dict(seq[k], kl_l=z4, kl_r=z4, ldesc_l=zd, ldesc_r=zd, oct_ll=zi, ang_l=np.zeros(0, np.float32))
    run_and_compare(oracle, [seq], cam, "kitti")


def test_seq_pipeline_many_sequences_copy_back_path(oracle):
    """More than 16 sequences: results and counts come back through explicit D2H copies instead of zero-copy writes."""
    cam = synth.KITTI_CAM
    seqs = [synth.make_stereo_sequence(1000 + b, n_frames=3, n_pts=120 + 7 * b, n_lines=12 if b % 3 else 0, cam=cam) for b in range(20)]
    run_and_compare(oracle, seqs, cam, "kitti", max_kp=512, max_kl=64)


def test_seq_pipeline_config5_eight_sequences_three_cameras(oracle):
    """BASELINE configs[4] on ONE GPU: KITTI sequences 00-07 side by side in one stvo_seq, each with the calibration and
    image size of its dataset (kitti00-02 / kitti03 / kitti04-10), points + lines, grid stereo + f2f + pose; every
    sequence must reproduce the oracle pipeline run with ITS camera."""
    seqs = [synth.make_config5_sequence(s, n_frames=4, n_pts=700 + 60 * s, n_lines=50 + 5 * s) for s in range(synth.CONFIG5_N_SEQUENCES)]
    run_and_compare(oracle, seqs, synth.CONFIG5_CAMS, "kitti")
    # the three calibrations really differ in what the pipeline computes: sequence 3 run with the camera of sequence 0 disagrees
    mp = match_params("kitti"); op = opt_params("kitti")
    a = pipeline_ref.run_sequence(oracle, seqs[3], synth.KITTI03_CAM, mp, op)
    b = pipeline_ref.run_sequence(oracle, seqs[3], synth.KITTI_CAM, mp, op)
    assert not np.allclose(a[0]["T"], b[0]["T"], atol=1e-6)


def test_seq_rotating_slots_equal_push(oracle):
    """stvo_seq_set_slots + upload / step_dev over more than two resident frames (what bench.py rotates through) gives
    exactly the results of pushing the same frames one by one."""
    from stvo_amd import capi
    cams = [synth.config5_cam(s) for s in (0, 3, 5)]
    seqs = [synth.make_config5_sequence(s, n_frames=5, n_pts=400, n_lines=40) for s in (0, 3, 5)]
    mp = match_params("kitti"); op = opt_params("kitti")
    ctx = capi.Context(device_id=0, max_rows=1024, max_batch=3)
    a = capi.Sequences(ctx, 3, 1024, 128, cams, mp, op)
    b = capi.Sequences(ctx, 3, 1024, 128, cams, mp, op)
    try:
        b.set_slots(5)
        for k in range(5):
            b.upload(k, [s[k] for s in seqs])
        order = [0, 1, 2, 3, 4, 3, 2, 1, 0, 1]   # ping-pong through the resident frames
        for k in order:
            ra, ca = a.push([s[k] for s in seqs])
            b.step_dev(k)
            rb, cb = b.read()
            assert np.array_equal(ca, cb)
            assert ra.tobytes() == rb.tobytes()   # bit-identical results
    finally:
        a.close(); b.close()
        ctx.close()


def test_seq_fetch_by_products(oracle):
    """stvo_seq_enable_fetch / fetch_matches / fetch_inliers (what the handler mirror rebuilds its host lists from): the raw
    stereo matches equal the oracle's grid matcher on the same frame, and the f2f matches / inlier flags are consistent
    with the pose result of the same step."""
    from stvo_amd import capi
    from stvo_amd.capi import StvoError
    cam = synth.KITTI_CAM
    seq = synth.make_stereo_sequence(1234, n_frames=3, n_pts=400, n_lines=40, cam=cam)
    mp = match_params("kitti"); op = opt_params("kitti")
    ctx = capi.Context(device_id=0, max_rows=2048, max_batch=1)
    dev = capi.Sequences(ctx, 1, 512, 64, cam, mp, op)
    try:
        with pytest.raises(StvoError):
            dev.fetch_matches()  # not enabled
        dev.enable_fetch(True)
        for k, fr in enumerate(seq):
            res, counts = dev.push([fr])
            ms_p, ms_l, m_p, m_l = dev.fetch_matches()
            ref = pipeline_ref.stereo_frame(oracle, fr, cam, mp, True, True)
            n_l, n_ll = len(fr["kp_l"]), len(fr["kl_l"])
            # the raw stereo matches (what matchGrid returns, stereoFrame.cpp:145,344) equal the oracle's, index for index
            assert np.array_equal(ms_p[0, :n_l], ref["m12_raw_p"])
            assert np.array_equal(ms_l[0, :n_ll], ref["m12_raw_l"])
            assert np.all(ms_p[0, n_l:] == -1) and np.all(ms_l[0, n_ll:] == -1)
            assert len(ref["P"]) == counts[0, 0] and len(ref["sP"]) == counts[0, 1]
            if k == 0:
                prev_counts = counts.copy()
                continue
            r = res[0]
            assert (m_p[0, :prev_counts[0, 0]] >= 0).sum() == r["n_matched_pt"]
            assert (m_l[0, :prev_counts[0, 1]] >= 0).sum() == r["n_matched_ls"]
            assert np.all(m_p[0, :prev_counts[0, 0]] < counts[0, 0])
            ip, il = dev.fetch_inliers()
            assert (ip[0, :prev_counts[0, 0]] == 1).sum() == r["n_inliers_pt"]
            assert (il[0, :prev_counts[0, 1]] == 1).sum() == r["n_inliers_ls"]
            assert np.array_equal(ip[0, :prev_counts[0, 0]] >= 0, m_p[0, :prev_counts[0, 0]] >= 0)
            prev_counts = counts.copy()
    finally:
        dev.close()
        ctx.close()


@pytest.mark.parametrize("env", [{}, {"STVO_SEQ_INLINE": "0"}], ids=["in-kernel", "events"])
@pytest.mark.parametrize("n_lines", [40, 0])
def test_seq_fetch_before_read_single_stream(oracle, switches, env, n_lines):
    """The StereoFrameHandler mirror's order of calls — upload, step, fetch_matches WHILE the pose kernel runs, then read: for one
    stream the pose kernel itself waits for the key-line stream and publishes the match indices (kernels.h: PoseArgs::wait_flag /
    fetch_*); STVO_SEQ_INLINE=0 keeps the events.  Same by-products and bit-identical results either way, equal to push + fetch."""
    from stvo_amd import capi
    switches(env)
    cam = synth.KITTI_CAM
    seq = synth.make_stereo_sequence(4242, n_frames=5, n_pts=900, n_lines=n_lines, cam=cam)
    mp = match_params("kitti"); op = opt_params("kitti", has_lines=1 if n_lines else 0)
    ctx = capi.Context(device_id=0, max_rows=2048, max_batch=1)
    a = capi.Sequences(ctx, 1, 2048, 64, cam, mp, op)
    b = capi.Sequences(ctx, 1, 2048, 64, cam, mp, op)
    try:
        a.enable_fetch(True); b.enable_fetch(True)
        for k, fr in enumerate(seq):
            a.upload(k & 1, [fr]); a.step_dev(k & 1)
            fa = a.fetch_matches()          # before the read: the pose kernel may still be running
            ra, ca = a.read()
            rb, cb = b.push([fr])
            fb = b.fetch_matches()
            ref = pipeline_ref.stereo_frame(oracle, fr, cam, mp, True, n_lines > 0)
            assert np.array_equal(fa[0][0, :len(fr["kp_l"])], ref["m12_raw_p"])
            for x, y in zip(fa, fb):
                assert np.array_equal(x, y)
            assert np.array_equal(ca, cb) and ra.tobytes() == rb.tobytes()
            if k > 0:
                ia, ib = a.fetch_inliers(), b.fetch_inliers()
                assert np.array_equal(ia[0], ib[0]) and np.array_equal(ia[1], ib[1])
                assert ra[0]["n_matched_pt"] > 100
    finally:
        a.close(); b.close()
        ctx.close()


@pytest.mark.parametrize("env", [{"STVO_GRID_FUSED": "0"}, {"STVO_GRID_FUSED_CAP": "-1"}, {"STVO_GRID_CELLS": "0"},
                                 {"STVO_GRID_CELLS": "0", "STVO_GRID_FUSED_CAP": "-1"}],
                         ids=["scan", "misfit", "cells-launch", "cells-launch-misfit"])
def test_seq_point_grid_other_formulations(oracle, switches, env):
    """The stereo point matcher runs as one workgroup per frame by default — for a batch this small it also builds the grid of
    its frame (point_cells.h); the scan formulation (separate launches), the scan formulation inside the fused launch (frames
    whose pairs do not fit the LDS) and point_cells_kernel as its own launch must give the same pipeline results."""
    switches(env)
    cam = synth.KITTI_CAM
    seqs = [synth.make_stereo_sequence(520 + b, n_frames=4, n_pts=500 + 500 * b, n_lines=30, cam=cam) for b in range(3)]
    run_and_compare(oracle, seqs, cam, "kitti")


@pytest.mark.parametrize("env", [
    {"STVO_LINE_FUSED": "1", "STVO_MATCH_SMALL": "1"},    # one workgroup per frame for 200 key-lines (LDS sized for 256, opt-in above 48 KB at 320)
    {"STVO_LINE_FUSED": "0", "STVO_MATCH_SMALL": "0"},    # the general grid matcher and match machinery for the key-lines
    {"STVO_LINE_FUSED": "1", "STVO_LINE_FORK": "late"},   # line stream forked after the point stage
    {"STVO_GRID_TAIL": "0"},                              # point_tail_kernel as its own launch behind the lean cells kernel
    {"STVO_MATCH_LAZY": "1"},                             # lazy reverse check for a tiny batch (default there: both directions in one scan)
], ids=["lines-fused", "lines-general", "late-fork", "tail-kernel", "lazy-reverse"])
def test_seq_step_variants_line_heavy(oracle, switches, env):
    """Every launch plan the step can choose (by batch size / line count, or by a developer switch) gives the oracle's results:
    EuRoC-shaped frames with ~240 key-lines per image, two streams."""
    switches(env)
    cam = synth.EUROC_CAM
    seqs = [synth.make_stereo_sequence(740 + b, n_frames=4, n_pts=500, n_lines=200, cam=cam, depth=(1.0, 8.0),
                                       octave_probs=[.5, .25, .15, .1], outlier_frac=0.2) for b in range(2)]
    run_and_compare(oracle, seqs, cam, "euroc", max_kl=512 if env.get("STVO_LINE_FORK") else 320)


def test_seq_batch_line_heavy_default_plan(oracle):
    """16 streams (the size from which the fused line kernel and the small-set match are the default for any line count) with
    150-260 key-lines per image and a stream without any."""
    cam = synth.EUROC_CAM
    seqs = [synth.make_stereo_sequence(7600 + b, n_frames=3, n_pts=260 + 10 * b, n_lines=0 if b == 5 else 130 + 6 * b, cam=cam, depth=(1.0, 8.0),
                                       octave_probs=[.5, .25, .15, .1], outlier_frac=0.2) for b in range(16)]
    run_and_compare(oracle, seqs, cam, "euroc")


def crowd(fr, rng, frac, box):
    """Moves a fraction of the key-points of both images into a small box (left / right keep their disparity): many candidates
    per window, long eligibility chains, near-duplicate descriptors."""
    fr = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in fr.items()}
    n = min(len(fr["kp_l"]), len(fr["kp_r"]))
    idx = rng.choice(n, int(frac * n), replace=False)
    u0, v0, w, h = box
    for key in ("kp_l", "kp_r"):
        kp = fr[key]
        kp[idx, 0] = (u0 + (kp[idx, 0] % w)).astype(np.float32)
        kp[idx, 1] = (v0 + (kp[idx, 1] % h)).astype(np.float32)
    # near-duplicate descriptors inside the crowd: ties and one-bit differences between candidates of the same window
    base = fr["desc_l"][idx[0]].copy()
    for j in idx[: len(idx) // 2]:
        flips = rng.integers(0, 256, size=rng.integers(0, 4))
        for side in ("desc_l", "desc_r"):
            d = base.copy()
            for f in flips:
                d[f >> 3] ^= np.uint8(1 << (f & 7))
            if side == "desc_r" and rng.random() < 0.5:
                d[0] ^= np.uint8(1)
            fr[side][j] = d
    return fr


@pytest.mark.parametrize("frac,box", [(0.5, (300, 100, 200, 12)), (0.95, (500, 200, 60, 7)), (0.3, (0, 0, 1241, 8)),
                                      (0.12, (200, 80, 600, 30)), (0.1, (200, 80, 500, 24))],
                         ids=["half-in-strip", "all-in-one-window", "top-row", "mild-crowd", "mild-crowd-wide-rows"])
@pytest.mark.parametrize("env", [{}, {"STVO_GRID_FUSED": "0"}, {"STVO_GRID_CELLS": "0"}], ids=["fused", "scan", "cells-launch"])
def test_seq_point_grid_crowded_frames(oracle, switches, frac, box, env):
    """Raw matchGrid output (stereoFrame.cpp:145) on frames whose key-points crowd into a few grid cells: rows with more than
    64 candidates and frames with more pairs than the one-workgroup formulation holds (it must then take the scan
    formulation on its own), chains of equal distances — index for index against the oracle."""
    from stvo_amd import capi
    switches(env)
    cam = synth.KITTI_CAM
    rng = np.random.default_rng(77)
    seq = synth.make_stereo_sequence(4321, n_frames=2, n_pts=1600, n_lines=0, cam=cam)
    mp = match_params("kitti"); op = opt_params("kitti", has_lines=0)
    ctx = capi.Context(device_id=0, max_rows=2048, max_batch=1)
    dev = capi.Sequences(ctx, 1, 2048, 64, cam, mp, op)
    try:
        dev.enable_fetch(True)
        for fr in seq:
            fr = crowd(fr, rng, frac, box)
            dev.push([fr])
            ms_p = dev.fetch_matches()[0]
            ref = pipeline_ref.stereo_frame(oracle, fr, cam, mp, True, False)
            n_l = len(fr["kp_l"])
            assert np.array_equal(ms_p[0, :n_l], ref["m12_raw_p"])
            assert (ref["m12_raw_p"] >= 0).sum() > 50
    finally:
        dev.close()
        ctx.close()


def test_seq_point_grid_persistent_workgroups_many_frames(oracle, switches):
    """More frames than CUs: every persistent workgroup of the one-workgroup-per-frame matcher takes several frames and
    prefetches the next one while it works — the raw stereo matches and the pose blocks of 600 small sequences must equal the
    scan formulation's, and sequence 0 / 599 the oracle's."""
    from stvo_amd import capi
    cam = synth.KITTI_CAM
    B = 600
    seqs = [synth.make_stereo_sequence(9000 + b, n_frames=2, n_pts=60 + (b % 7) * 40, n_lines=0, cam=cam) for b in range(B)]
    mp = match_params("kitti"); op = opt_params("kitti", has_lines=0)

    def run():
        ctx = capi.Context(device_id=0, max_rows=512, max_batch=B)
        dev = capi.Sequences(ctx, B, 512, 64, cam, mp, op)
        try:
            dev.enable_fetch(True)
            out = []
            for k in range(2):
                res, counts = dev.push([seqs[b][k] for b in range(B)])
                ms_p = dev.fetch_matches()[0].copy()
                out.append((ms_p, counts.copy(), res.copy() if k else None))
            return out
        finally:
            dev.close()
            ctx.close()

    fused = run()
    switches({"STVO_GRID_FUSED": "0"})
    scan = run()
    for k in range(2):
        assert np.array_equal(fused[k][0], scan[k][0]) and np.array_equal(fused[k][1], scan[k][1])
    assert np.array_equal(fused[1][2]["T"], scan[1][2]["T"]) and np.array_equal(fused[1][2]["iters"], scan[1][2]["iters"])
    assert np.array_equal(fused[1][2]["status"], scan[1][2]["status"])
    for b in (0, B - 1):
        ref = pipeline_ref.stereo_frame(oracle, seqs[b][1], cam, mp, True, False)
        assert np.array_equal(fused[1][0][b, :len(seqs[b][1]["kp_l"])], ref["m12_raw_p"])


@pytest.mark.parametrize("pose", ["default", "1", "4:2", "4:4"])
def test_seq_pipeline_headline_shape_every_stream_vs_oracle(oracle, switches, pose):
    """The shape bench.py's `value` is quoted on — hundreds of streams x (1650 landmarks ~ 2000 key-points + 85 segments ~ 100
    key-lines), the eight sequence ids / three KITTI calibrations of configs[4], resident frame slots advanced with
    upload / step_dev in ping-pong order, the batch-size default of the pose kernel — with EVERY stream compared with the
    oracle-driven per-frame loop on every transition (forward and backward), not with another formulation of itself."""
    from concurrent.futures import ThreadPoolExecutor
    from stvo_amd import capi
    if pose != "default":   # "1": pose_kernel.hip (worker waves + solver wave), "4:n": pose_kernel2p.hip with n waves per frame pair
        k, _, nw = pose.partition(":")
        switches(dict({"STVO_POSE_KERNEL": k}, **({"STVO_POSE2P_NW": nw} if nw else {})))
    B, S = 320, 3
    ids = np.arange(B) % synth.CONFIG5_N_SEQUENCES
    streams = [synth.make_config5_sequence(int(s), n_frames=S, n_pts=1650, n_lines=85, replica=400 + b // 8) for b, s in enumerate(ids)]
    cams = [synth.config5_cam(int(s)) for s in ids]
    mp = match_params("kitti"); op = opt_params("kitti")
    order = [0, 1, 2, 1, 0]   # 0->1, 1->2 forward; 2->1, 1->0 backward

    def ref_pair(args):
        b, a, c = args
        return pipeline_ref.run_sequence(oracle, [streams[b][a], streams[b][c]], cams[b], mp, op)[0]

    ctx = capi.Context(device_id=0, max_rows=2048, max_batch=B)
    dev = capi.Sequences(ctx, B, 2048, 128, cams, mp, op)
    try:
        dev.set_slots(S)
        for k in range(S):
            dev.upload(k, [st[k] for st in streams])
        prev = None
        with ThreadPoolExecutor(16) as ex:   # the oracle's C functions run outside the GIL
            for cur in order:
                dev.step_dev(cur)
                res, counts = dev.read()
                if prev is not None:
                    refs = list(ex.map(ref_pair, [(b, prev, cur) for b in range(B)]))
                    for b, o in enumerate(refs):
                        r = res[b]
                        assert counts[b, 0] == o["n_stereo_pt"] and counts[b, 1] == o["n_stereo_ls"], (b, prev, cur)
                        assert r["n_matched_pt"] == o["n_matched_pt"] and r["n_matched_ls"] == o["n_matched_ls"], (b, prev, cur)
                        assert r["status"] == o["status"] and r["path"] == o["path"] and tuple(r["iters"]) == o["iters"], (b, prev, cur)
                        assert r["n_inliers_pt"] == o["n_inliers_pt"] and r["n_inliers_ls"] == o["n_inliers_ls"], (b, prev, cur)
                        T = r["T"].reshape(4, 4)
                        assert np_model.rot_angle(T[:3, :3], o["T"][:3, :3]) < 1e-4 and np.linalg.norm(T[:3, 3] - o["T"][:3, 3]) < 1e-3
                        assert np.allclose(T, o["T"], atol=1e-8), (b, prev, cur)
                    assert (res["status"] == 0).mean() > 0.95 and res["n_matched_pt"].mean() > 1300
                prev = cur
    finally:
        dev.close()
        ctx.close()


def test_seq_steps_back_to_back_schedules_agree(switches):
    """Round 6: three stereo sets, two copies of the f2f match indices, and two schedules built on them — the key-line stage one step
    AHEAD (the line stream waits for the previous step's fork event and a gate behind the dispatch of the previous pose kernel: the
    default for > 2 x CUs streams with ~100 key-lines per image) and the pipelined steps (optimizePose on the aux stream beside the next
    step's stereo association, the persistent point matcher taking frames by ticket: opt-in).  Seven steps enqueued BACK TO BACK — only
    then do the overlaps they are about happen — must leave exactly the poses, counts and inlier totals of the plain schedule with a read
    after every step (the one the headline-shape test compares with the oracle stream by stream)."""
    from stvo_amd import capi
    B, S = 320, 3
    ids = np.arange(B) % synth.CONFIG5_N_SEQUENCES
    streams = [synth.make_config5_sequence(int(s), n_frames=S, n_pts=900, n_lines=60, replica=700 + b // 8) for b, s in enumerate(ids)]
    cams = [synth.config5_cam(int(s)) for s in ids]
    mp = match_params("kitti"); op = opt_params("kitti")
    order = [0, 1, 2, 1, 0, 1, 2]

    def run(env, read_every_step):
        switches(env)
        ctx = capi.Context(device_id=0, max_rows=2048, max_batch=B)
        dev = capi.Sequences(ctx, B, 2048, 128, cams, mp, op)
        try:
            dev.set_slots(S)
            for k in range(S):
                dev.upload(k, [st[k] for st in streams])
            for cur in order:
                dev.step_dev(cur)
                if read_every_step:
                    dev.read()
            res, counts = dev.read()
            return res.copy(), counts.copy()
        finally:
            dev.close()
            ctx.close()

    base = {"STVO_LINES_AHEAD": "0", "STVO_SEQ_PIPE": "0", "STVO_GRID_DYN": "1"}
    ref_res, ref_counts = run(base, True)
    assert (ref_res["status"] == 0).mean() > 0.9 and ref_counts[:, 2].mean() > 500
    for name, env in (("plain, back to back", base), ("key-line stage ahead", dict(base, STVO_LINES_AHEAD="1")),
                      ("key-line stage ahead, gate in front of the cells kernel", dict(base, STVO_LINES_AHEAD="2")),
                      ("pipelined steps", dict(base, STVO_SEQ_PIPE="1")), ("pipelined steps, static frames", dict(base, STVO_SEQ_PIPE="1", STVO_GRID_DYN="0")),
                      ("pipelined steps, no gate", dict(base, STVO_SEQ_PIPE="2"))):
        res, counts = run(env, False)
        assert np.array_equal(counts, ref_counts), name
        assert res.tobytes() == ref_res.tobytes(), name
