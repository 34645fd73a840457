#!/usr/bin/env python3
"""Throughput of the WHOLE device-resident per-frame pipeline (stvo_seq_*) on one MI355X — BASELINE configs[2]
shape: KITTI-00-shaped stereo with points + lines, grid-windowed stereo match + f2f match + full GN pose, for B
independent sequences in lock-step, features resident in HBM (two frame slots per sequence, alternated).
Secondary measurement (the headline metric of bench.py is configs[1]); prints one JSON line.

    python tools/bench_pipeline.py [--batch 256] [--steps 20] [--points 1650] [--lines 85]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--points", type=int, default=1650)  # + 20 % distractors ~ 2000 key-points per image
    ap.add_argument("--lines", type=int, default=85)     # + 20 % ~ 100 key-lines per image
    ap.add_argument("--cpu-frames", type=int, default=8, help="oracle sample: this many 9-frame sequences (0: skip)")
    ap.add_argument("--preset", default="kitti", choices=["kitti", "euroc"],
                    help="euroc: BASELINE configs[3] shape (752x480, 800 points over 4 octaves, 300 lines, 40 %% outliers)")
    ap.add_argument("--mode", type=int, default=0, help="0 GN, 1 robust GN, 2 LM")
    ap.add_argument("--no-point-stage", action="store_true", help="has_points = 0: only the key-line kernels run (their standalone durations under rocprofv3)")
    a = ap.parse_args()
    import torch  # noqa: F401  (one HIP runtime per process, see capi.load)
    from stvo_amd import capi, synth
    from stvo_amd.ctypes_types import match_params, opt_params
    euroc = a.preset == "euroc"
    cam = synth.EUROC_CAM if euroc else synth.KITTI_CAM
    mp = match_params(a.preset)
    op = opt_params(a.preset, has_lines=1 if a.lines > 0 else 0, mode=a.mode, has_points=0 if a.no_point_stage else 1)
    B = a.batch
    extra = dict(depth=(0.5, 8.0), octave_probs=[.5, .25, .15, .1], outlier_frac=0.4) if euroc else {}
    seqs = [synth.make_stereo_sequence(synth.frame_seed(b, 0), n_frames=2, n_pts=a.points, n_lines=a.lines, cam=cam, **extra)
            for b in range(B)]
    ctx = capi.Context(device_id=0, max_rows=2048, max_batch=B)
    dev = capi.Sequences(ctx, B, 2048, 512, cam, mp, op)
    dev.upload(0, [s[0] for s in seqs])
    dev.upload(1, [s[1] for s in seqs])
    for k in range(a.warmup):
        dev.step_dev(k & 1)
    ctx.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        dev.step_dev((a.warmup + k) & 1)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    res, counts = dev.read()
    ok = float((res["status"] == 0).mean())
    out = {"metric": "stereo frames/s (grid stereo match + f2f match + optimizePose), device-resident pipeline",
           "value": B * a.steps / dt, "unit": "frame-pairs/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": dt / a.steps * 1e3, "data": "synthetic",
           "config": {"workload": ("BASELINE configs[3]: EuRoC-shaped 752x480 stereo, line-heavy, 40 % point outliers, "
                                   f"optimiser mode {a.mode} (0 GN / 1 robust GN / 2 LM)" if euroc else
                                   "BASELINE configs[2]: KITTI-00-shaped stereo, points + lines, grid-windowed stereo association, "
                                   "f2f mutual matching, GN pose") + "; B independent sequences, two frames per sequence resident in HBM",
                      "sequences": B, "keypoints_per_image": len(seqs[0][0]["kp_l"]), "keylines_per_image": len(seqs[0][0]["kl_l"]),
                      "mean_stereo_points": float(counts[:, 0].mean()), "mean_stereo_lines": float(counts[:, 1].mean()),
                      "mean_matched_points": float(counts[:, 2].mean()), "committed_pose_fraction": ok}}
    if a.cpu_frames > 0:
        # the oracle on the same per-frame pipeline: sequences of 9 frames (9 stereo associations, 8 x (f2f + optimizePose)), one
        # untimed sequence first (library load, page faults), rate = frames / time
        import oracle_lib
        import pipeline_ref
        orc = oracle_lib.load()
        cpu_seqs = [synth.make_stereo_sequence(synth.frame_seed(1000 + b, 0), n_frames=9, n_pts=a.points, n_lines=a.lines, cam=cam, **extra)
                    for b in range(a.cpu_frames + 1)]
        pipeline_ref.run_sequence(orc, cpu_seqs[0], cam, mp, op)
        t1 = time.perf_counter()
        for sq in cpu_seqs[1:]:
            pipeline_ref.run_sequence(orc, sq, cam, mp, op)
        dtc = time.perf_counter() - t1
        nfr = 9 * a.cpu_frames
        out["cpu_baseline"] = {"value": nfr / dtc, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
                               "sample": f"{a.cpu_frames} sequences x 9 frames (stereo association + f2f + optimizePose per frame), oracle, {dtc:.1f} s"}
    dev.close()
    ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
