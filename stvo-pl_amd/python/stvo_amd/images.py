"""Images in, poses out, nothing through the host: the ORB point front-end (stvo_orb_detect_dev) feeding the device-resident
per-frame pipeline (stvo_seq_upload_dev + stvo_seq_step_dev) for B stereo streams.  Plumbing for tests and bench only: torch
owns the device buffers, every computation is behind the C-ABI.

Replaces, per frame and stream: StereoFrame::detectStereoPoints + matchStereoPoints (/root/reference/src/stereoFrame.cpp:88-173),
StereoFrameHandler::f2fTracking + optimizePose (src/stereoFrameHandler.cpp:106-392) — key-points only (the LSD / LBD line
front-end is not built)."""
import ctypes as C

import numpy as np
import torch

from . import capi
from .capi import FrameFeatures


class ImagePipeline:
    def __init__(self, ctx, B, cam, mp, op, max_kp=2048, nfeatures=2000, fast_threshold=20, edge_threshold=19, device="cuda:0", nlevels=1,
                 scale_factor=1.2):
        """cam: one camera dict (width / height = image size) for all B streams.  op must have has_lines = 0.  nlevels / scale_factor:
        Config::orbNLevels / orbScaleFactor (the key-point octaves travel with the key-points: sigma2 = 1 / scale^(2 level))."""
        self.ctx, self.B, self.K = ctx, B, max_kp
        self.cols, self.rows = cam["width"], cam["height"]
        self.orb = capi.Orb(ctx, 2 * B, self.cols, self.rows, max_kp, nfeatures, fast_threshold, edge_threshold, nlevels, scale_factor)  # left images, then right
        self.seq = capi.Sequences(ctx, B, max_kp, 64, cam, mp, op)
        dev = torch.device(device)
        self.img = torch.zeros((2 * B, self.rows, self.cols), dtype=torch.uint8, device=dev)
        self.kp = torch.zeros((2 * B, max_kp, 2), dtype=torch.float32, device=dev)
        self.resp = torch.zeros((2 * B, max_kp), dtype=torch.float32, device=dev)
        self.ang = torch.zeros((2 * B, max_kp), dtype=torch.float32, device=dev)
        self.desc = torch.zeros((2 * B, max_kp, 32), dtype=torch.uint8, device=dev)
        self.n = torch.zeros((2 * B,), dtype=torch.int32, device=dev)
        self.oct = torch.zeros((2 * B, max_kp), dtype=torch.int32, device=dev)
        ff = FrameFeatures()
        ff.stride_kp, ff.stride_kl = max_kp, 0
        ff.n_kp_l = C.c_void_p(self.n.data_ptr())
        ff.n_kp_r = C.c_void_p(self.n.data_ptr() + 4 * B)
        ff.kp_l = C.c_void_p(self.kp.data_ptr())
        ff.kp_r = C.c_void_p(self.kp.data_ptr() + 8 * B * max_kp)
        ff.desc_l = C.c_void_p(self.desc.data_ptr())
        ff.desc_r = C.c_void_p(self.desc.data_ptr() + 32 * B * max_kp)
        ff.oct_l = C.c_void_p(self.oct.data_ptr())
        self.ff = ff  # every line pointer stays NULL: no key-lines
        self.slot = 0

    def set_images(self, left, right):
        """left / right: uint8 [B, rows, cols] (numpy or torch); copied into the resident image buffer."""
        B = self.B
        self.img[:B].copy_(torch.as_tensor(left).reshape(B, self.rows, self.cols), non_blocking=True)
        self.img[B:].copy_(torch.as_tensor(right).reshape(B, self.rows, self.cols), non_blocking=True)

    def enqueue(self, img_ptr=None):
        """Detection + description of the 2 B resident images (or of the uint8 [2 B, rows, cols] device buffer at img_ptr: B left,
        then B right), ingestion, one pipeline step — all asynchronous."""
        self.orb.detect_dev(img_ptr if img_ptr is not None else self.img.data_ptr(), self.kp.data_ptr(), self.resp.data_ptr(), self.ang.data_ptr(), self.desc.data_ptr(),
                            self.n.data_ptr(), octave=self.oct.data_ptr())
        self.seq.upload_dev(self.slot, self.ff)
        self.seq.step_dev(self.slot)
        self.slot ^= 1

    def push_images(self, left, right):
        """One frame of every stream: (pose results [B], counts [B, 4]) like Sequences.push."""
        self.set_images(left, right)
        torch.cuda.current_stream().synchronize()  # the copies above ran on torch's stream, the library uses its own
        self.enqueue()
        return self.seq.read()

    def close(self):
        self.seq.close()
        self.orb.close()
