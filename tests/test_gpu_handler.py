"""End-to-end drop-in check: the C++ StVO::StereoFrameHandler mirror (stvo-pl_amd/host, driven by
app/imagesStVO_synth like the reference's imagesStVO.cpp loop) on the GPU vs the oracle-driven pipeline,
frame by frame: stereo association (grid matchers + filters), f2f tracking, optimizePose, Tfw
composition, adaptive FAST.  Tolerance: 1e-4 rad / 1e-3 m per frame (BASELINE.json); counts exact."""
import os
import subprocess

import numpy as np
import pytest

import np_model
import pipeline_ref
from stvo_amd import synth
from stvo_amd.ctypes_types import match_params, opt_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "stvo-pl_amd", "bin", "imagesStVO_synth")


def run_app(tmp_path, frames, cam, preset, mode=0, extra=(), pipeline=True):
    """pipeline=True: the handler runs on the device-resident pipeline (its default); False: one synchronous C-ABI
    call per stage (STVO_HANDLER_PIPELINE=0)."""
    seq = str(tmp_path / "seq.bin"); res = str(tmp_path / "res.bin")
    synth.write_sequence(seq, frames, cam)
    env = dict(os.environ, STVO_HANDLER_PIPELINE="1" if pipeline else "0")
    p = subprocess.run([APP, seq, res, "--preset", preset, "--mode", str(mode), *extra], capture_output=True, text=True, timeout=300,
                       env=env)
    assert p.returncode == 0, p.stderr + p.stdout
    return synth.read_results(res), p.stdout


def compare(res, ref):
    assert len(res) == len(ref)
    for k, (r, o) in enumerate(zip(res, ref)):
        ints = r["ints"]
        assert ints[1] == o["status"] and ints[2] == o["path"], (k, ints, o["status"], o["path"])
        assert (ints[3], ints[4]) == o["iters"]
        assert ints[5] == o["n_matched_pt"] and ints[6] == o["n_inliers_pt"]
        assert ints[7] == o["n_matched_ls"] and ints[8] == o["n_inliers_ls"]
        assert ints[9] == o["n_stereo_pt"] and ints[10] == o["n_stereo_ls"]
        DT = r["DT"].reshape(4, 4); Tfw = r["Tfw"].reshape(4, 4)
        assert np_model.rot_angle(DT[:3, :3], o["T"][:3, :3]) < 1e-4 and np.linalg.norm(DT[:3, 3] - o["T"][:3, 3]) < 1e-3
        assert np.allclose(DT, o["T"], atol=1e-8)
        assert np.allclose(r["DT_cov"].reshape(6, 6), o["cov"], rtol=1e-6, atol=1e-12)
        assert np.isclose(r["err"], o["err"], rtol=1e-8)
        assert np.allclose(Tfw, o["Tfw"], atol=1e-7)
        assert np.allclose(r["Tfw_cov"].reshape(6, 6), o["Tfw_cov"], rtol=1e-6, atol=1e-10)
        assert r["fast"] == o["fast"]


@pytest.mark.parametrize("pipeline", [True, False])
def test_kitti_points_and_lines_sequence(tmp_path, oracle, pipeline):
    cam = synth.KITTI_CAM
    frames = synth.make_stereo_sequence(2025, n_frames=6, n_pts=700, n_lines=70, cam=cam)
    res, out = run_app(tmp_path, frames, cam, "kitti", pipeline=pipeline)
    ref = pipeline_ref.run_sequence(oracle, frames, cam, match_params("kitti"), opt_params("kitti"))
    compare(res, ref)
    assert all(r["ints"][1] == 0 for r in res)            # every frame committed
    assert "Proc. time" in out and "Points:" in out and "Lines:" in out
    # and the estimate follows the true motion
    for r, fr in zip(res, frames[1:]):
        Tt = np.linalg.inv(fr["T_true"])
        DT = r["DT"].reshape(4, 4)
        assert np_model.rot_angle(DT[:3, :3], Tt[:3, :3]) < 5e-3 and np.linalg.norm(DT[:3, 3] - Tt[:3, 3]) < 0.1


@pytest.mark.parametrize("pipeline", [True, False])
def test_kitti_points_only_2000(tmp_path, oracle, pipeline):
    """BASELINE configs[0] shape: KITTI-00-like pairs, points only, through the handler API."""
    cam = synth.KITTI_CAM
    frames = synth.make_stereo_sequence(7, n_frames=4, n_pts=1650, n_lines=0, cam=cam)  # 1650 + 20% = ~2000 key-points
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text("has_lines : false   # points-only\n")
    res, _ = run_app(tmp_path, frames, cam, "kitti", extra=("-c", str(cfg)), pipeline=pipeline)
    ref = pipeline_ref.run_sequence(oracle, frames, cam, match_params("kitti"), opt_params("kitti", has_lines=0))
    compare(res, ref)


@pytest.mark.parametrize("mode,pipeline", [(0, True), (1, True), (2, True), (0, False), (2, False)])
def test_euroc_shaped_line_heavy(tmp_path, oracle, mode, pipeline):
    cam = synth.EUROC_CAM
    frames = synth.make_stereo_sequence(99 + mode, n_frames=4, n_pts=500, n_lines=200, cam=cam, depth=(1.0, 8.0),
                                        octave_probs=[.5, .25, .15, .1], outlier_frac=0.2)
    res, _ = run_app(tmp_path, frames, cam, "euroc", mode=mode, pipeline=pipeline)
    fast = dict(adaptive=True, th0=20, mn=5, mx=50, inc=5, feat=50, err=0.5)
    ref = pipeline_ref.run_sequence(oracle, frames, cam, match_params("euroc"), opt_params("euroc", mode=mode), fast)
    compare(res, ref)


def test_oversized_frame_falls_back_to_the_per_call_path(tmp_path, oracle):
    """A frame with more key-points than the device pipeline holds (2048 per image) makes the handler continue on the
    per-call path from that frame on, without losing the sequence."""
    cam = synth.KITTI_CAM
    frames = synth.make_stereo_sequence(31, n_frames=5, n_pts=900, n_lines=0, cam=cam)
    big = synth.make_stereo_sequence(32, n_frames=1, n_pts=2400, n_lines=0, cam=cam)[0]   # ~2900 key-points per image
    frames[3] = dict(big, T_true=frames[3]["T_true"])
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text("has_lines : false\n")
    res, _ = run_app(tmp_path, frames, cam, "kitti", extra=("-c", str(cfg)))
    ref = pipeline_ref.run_sequence(oracle, frames, cam, match_params("kitti"), opt_params("kitti", has_lines=0))
    compare(res, ref)


@pytest.mark.parametrize("pipeline", [True, False])
def test_motion_model_through_the_handler(tmp_path, oracle, pipeline):
    """use_motion_model : true (Config::useMotionModel, src/stereoFrameHandler.cpp:317-324): the initial DT of optimizePose is the
    increment committed for the previous pair unless !isGoodSolution(prev) — on the device-resident pipeline the rule is applied ON
    the device by the previous step's commit (stvo_seq_set_motion_model), on the per-call path by the handler; both against the
    oracle pipeline with the same rule.  One frame without right-camera features makes two pairs fail: the pair after a rejected one
    starts from the identity again."""
    cam = synth.KITTI_CAM
    frames = synth.make_stereo_sequence(3131, n_frames=9, n_pts=600, n_lines=50, cam=cam)
    z2 = np.zeros((0, 2), np.float32); zd = np.zeros((0, 32), np.uint8); z4 = np.zeros((0, 4), np.float32)
    frames[4] = dict(frames[4], kp_r=z2, desc_r=zd, kl_r=z4, ldesc_r=zd)
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text("use_motion_model : true\n")
    res, _ = run_app(tmp_path, frames, cam, "kitti", extra=("-c", str(cfg)), pipeline=pipeline)
    ref = pipeline_ref.run_sequence(oracle, frames, cam, match_params("kitti"), opt_params("kitti"), motion_model=True)
    compare(res, ref)
    assert [r["ints"][1] == 0 for r in res] == [True, True, True, False, False, True, True, True]
    plain = pipeline_ref.run_sequence(oracle, frames, cam, match_params("kitti"), opt_params("kitti"))
    assert any(a["iters"] != b["iters"] for a, b in zip(ref, plain))   # the prior changes the optimisation's course


def test_keyframe_decisions_and_cli_offset_step(tmp_path, oracle):
    """--keyframes: needNewKF / currFrameIsKF (src/stereoFrameHandler.cpp:1136-1218) after every optimizePose of the handler,
    against the oracle's restatement — same decisions, and the poses after a key-frame are expressed in the restarted map
    frame.  -o / -s: the reference's frame offset / step options (app/imagesStVO.cpp:138-171)."""
    cam = synth.KITTI_CAM
    frames = synth.make_stereo_sequence(4242, n_frames=16, n_pts=500, n_lines=40, cam=cam)
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text("max_kf_t_dist : 2.5   # a key-frame every few metres of forward motion\n")
    res, _ = run_app(tmp_path, frames, cam, "kitti", extra=("--keyframes", "-c", str(cfg)))
    ref = pipeline_ref.run_sequence(oracle, frames, cam, match_params("kitti"), opt_params("kitti"),
                                    keyframes=dict(min_entropy_ratio=0.85, max_kf_t_dist=2.5, max_kf_r_dist=15.0))
    compare(res, ref)
    got = [int(r["pad"]) for r in res]
    assert got == [o["new_kf"] for o in ref]
    assert 2 <= sum(got) < len(got)                      # both outcomes occur
    # offset 3, every second frame, 5 frames: frames 3, 5, 7, 9, 11
    res2, _ = run_app(tmp_path, frames, cam, "kitti", extra=("-o", "3", "-s", "2", "-n", "5"))
    sub = [frames[k] for k in (3, 5, 7, 9, 11)]
    ref2 = pipeline_ref.run_sequence(oracle, sub, cam, match_params("kitti"), opt_params("kitti"))
    compare(res2, ref2)


@pytest.mark.parametrize("preset,nlevels", [("kitti", 1), ("euroc", 4)])
def test_image_entry_points_of_the_handler(tmp_path, oracle, preset, nlevels):
    """initialize / insertStereoPair(img_l, img_r, idx) (include/stereoFrameHandler.h:44-45) through the C++ mirror: stereo images in,
    the ORB point front-end on the GPU with Config's orb_* values (1 level for config_kitti.yaml, 4 at 1.2 for config_euroc.yaml) and
    the handler's ADAPTIVE FAST threshold, then the usual path — against the CPU chain: ORB oracle on every image with the threshold
    the previous frame left behind, its key-points through the oracle-driven per-frame loop."""
    cam = dict(synth.KITTI_CAM if preset == "kitti" else synth.EUROC_CAM, width=640, height=240)
    pairs = synth.make_stereo_image_sequence(77, 5, cam)
    seq = str(tmp_path / "img.bin"); res_path = str(tmp_path / "res.bin")
    synth.write_image_sequence(seq, pairs, cam)
    p = subprocess.run([APP, seq, res_path, "--preset", preset, "--no-lines"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr + p.stdout
    res = synth.read_results(res_path)
    mp = match_params(preset); op = opt_params(preset, has_lines=0)
    nfeat = {"kitti": 2000, "euroc": 800}[preset]
    fast = dict(adaptive=True, th0=20, mn=7, mx=30, inc=5, feat=50, err=0.5) if preset == "kitti" else \
        dict(adaptive=True, th0=20, mn=5, mx=50, inc=5, feat=50, err=0.5)   # config_kitti.yaml / config_euroc.yaml fast_* values
    pattern = oracle.orb_default_pattern()
    z4 = np.zeros((0, 4), np.float32); zd = np.zeros((0, 32), np.uint8)
    frames, th, ref = [], fast["th0"], []
    for k, (left, right) in enumerate(pairs):
        l = oracle.orb_detect_levels(left, nfeatures=nfeat, nlevels=nlevels, fast_th=th, pattern=pattern)
        r = oracle.orb_detect_levels(right, nfeatures=nfeat, nlevels=nlevels, fast_th=th, pattern=pattern)
        frames.append(dict(kp_l=l["kp"], oct_l=l["octave"], desc_l=l["desc"], kp_r=r["kp"], desc_r=r["desc"], kl_l=z4, oct_ll=np.zeros(0, np.int32),
                           ldesc_l=zd, kl_r=z4, ldesc_r=zd, ang_l=np.zeros(0, np.float32)))
        if k:
            ref = pipeline_ref.run_sequence(oracle, frames, cam, mp, op, fast=fast)
            th = ref[-1]["fast"]   # updateFrame's threshold for the NEXT detection
    compare(res, ref)
    assert sum(r["ints"][1] == 0 for r in res) >= 2 and "FAST:" in p.stdout


@pytest.mark.parametrize("preset,nlevels,refine", [("kitti", 1, 0), ("euroc", 4, 0), ("kitti", 1, 1)])
def test_image_entry_points_points_and_lines(tmp_path, oracle, preset, nlevels, refine):
    """The same with Config::hasLines(): insertStereoPair(img_l, img_r, idx) detects key-points (ORB) AND key-lines — the LSD detector
    with Config's lsd_* options and min_line_length x min(cols, rows), the top-N cut by response, LBD descriptors
    (src/stereoFrame.cpp:191-243) — on the GPU and runs points + lines through the usual path; against the CPU chain: ORB / LSD / LBD
    oracles on every image, their features through the oracle-driven per-frame loop."""
    cam = dict(synth.KITTI_CAM if preset == "kitti" else synth.EUROC_CAM, width=640, height=240)
    pairs = synth.make_stereo_image_sequence(91, 4, cam)
    seq = str(tmp_path / "img.bin"); res_path = str(tmp_path / "res.bin")
    synth.write_image_sequence(seq, pairs, cam)
    extra = ()
    if refine:   # lsd_refine : 1 (src/config.cpp:105,198) travels through Config to the detector (LSD_REFINE_STD)
        cfg = tmp_path / "cfg.yaml"
        cfg.write_text(f"lsd_refine : {refine}\n")
        extra = ("-c", str(cfg))
    p = subprocess.run([APP, seq, res_path, "--preset", preset, *extra], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr + p.stdout
    res = synth.read_results(res_path)
    mp = match_params(preset); op = opt_params(preset, has_lines=1)
    nfeat = {"kitti": 2000, "euroc": 800}[preset]; nlines = {"kitti": 100, "euroc": 300}[preset]
    fast = dict(adaptive=True, th0=20, mn=7, mx=30, inc=5, feat=50, err=0.5) if preset == "kitti" else \
        dict(adaptive=True, th0=20, mn=5, mx=50, inc=5, feat=50, err=0.5)
    pattern = oracle.orb_default_pattern()
    lopts = oracle.lsd_opts(min_length=0.025 * min(cam["width"], cam["height"]), nfeatures=nlines, refine=refine)

    def lines_of(img):
        kl = oracle.lsd_detect(img, lopts)
        rec = np.stack([kl["sx"], kl["sy"], kl["ex"], kl["ey"], kl["angle"]], axis=1).astype(np.float32)
        desc = oracle.lbd_compute(img, rec, kl["num_pixels"])
        return np.ascontiguousarray(rec[:, :4]), np.ascontiguousarray(rec[:, 4]), desc

    frames, th, ref = [], fast["th0"], []
    for k, (left, right) in enumerate(pairs):
        l = oracle.orb_detect_levels(left, nfeatures=nfeat, nlevels=nlevels, fast_th=th, pattern=pattern)
        r = oracle.orb_detect_levels(right, nfeatures=nfeat, nlevels=nlevels, fast_th=th, pattern=pattern)
        kl_l, ang_l, ld_l = lines_of(left)
        kl_r, _, ld_r = lines_of(right)
        assert len(kl_l) > 20 and len(kl_r) > 20
        frames.append(dict(kp_l=l["kp"], oct_l=l["octave"], desc_l=l["desc"], kp_r=r["kp"], desc_r=r["desc"], kl_l=kl_l,
                           oct_ll=np.zeros(len(kl_l), np.int32), ldesc_l=ld_l, kl_r=kl_r, ldesc_r=ld_r, ang_l=ang_l))
        if k:
            ref = pipeline_ref.run_sequence(oracle, frames, cam, mp, op, fast=fast)
            th = ref[-1]["fast"]
    compare(res, ref)
    assert any(r["ints"][7] > 0 for r in res[1:])  # matched key-lines took part
    if refine:   # ... and the mode changes the key-lines of these images (the yaml key was not ignored on either side)
        plain = oracle.lsd_detect(pairs[0][0], oracle.lsd_opts(min_length=0.025 * min(cam["width"], cam["height"]), nfeatures=nlines))
        mine = oracle.lsd_detect(pairs[0][0], lopts)
        assert not all(np.array_equal(plain[f], mine[f]) for f in ("sx", "sy", "ex", "ey"))
