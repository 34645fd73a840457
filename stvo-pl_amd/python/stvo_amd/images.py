"""Images in, poses out, nothing through the host: the ORB point front-end (stvo_orb_detect_dev) feeding the device-resident
per-frame pipeline (stvo_seq_upload_dev + stvo_seq_step_dev) for B stereo streams.  Plumbing for tests and bench only: torch
owns the device buffers, every computation is behind the C-ABI.

Replaces, per frame and stream: StereoFrame::detectStereoPoints + matchStereoPoints (/root/reference/src/stereoFrame.cpp:88-173),
StereoFrameHandler::f2fTracking + optimizePose (src/stereoFrameHandler.cpp:106-392); with lsd = stvo_lsd_params also
detectStereoLineSegments + matchStereoLines (:191-243, :309-398): LSD detector, top-N cut, LBD descriptors (stvo_lsd_* / stvo_lbd_*)."""
import ctypes as C

import numpy as np
import torch

from . import capi
from .capi import FrameFeatures


class ImagePipeline:
    def __init__(self, ctx, B, cam, mp, op, max_kp=2048, nfeatures=2000, fast_threshold=20, edge_threshold=19, device="cuda:0", nlevels=1,
                 scale_factor=1.2, lsd=None, max_kl=128):
        """cam: one camera dict (width / height = image size) for all B streams.  nlevels / scale_factor: Config::orbNLevels /
        orbScaleFactor (the key-point octaves travel with the key-points: sigma2 = 1 / scale^(2 level)).  lsd: capi.lsd_params(...)
        for the key-line front-end (op.has_lines = 1, at most max_kl key-lines per image), None: key-points only (op.has_lines = 0)."""
        self.ctx, self.B, self.K, self.M = ctx, B, max_kp, max_kl
        self.cols, self.rows = cam["width"], cam["height"]
        self.orb = capi.Orb(ctx, 2 * B, self.cols, self.rows, max_kp, nfeatures, fast_threshold, edge_threshold, nlevels, scale_factor)  # left images, then right
        if lsd is not None and (lsd.nfeatures == 0 or lsd.nfeatures > max_kl):
            import warnings
            warnings.warn(f"ImagePipeline: lsd nfeatures = {lsd.nfeatures} against a capacity of {max_kl} key-lines per image: the strongest {max_kl} are kept")
        self.lsd = capi.Lsd(ctx, 2 * B, self.cols, self.rows, lsd, max_keylines=max_kl) if lsd is not None else None
        self.lbd = capi.Lbd(ctx, 2 * B, self.cols, self.rows, max_keylines=max_kl) if lsd is not None else None
        self.seq = capi.Sequences(ctx, B, max_kp, max_kl if lsd is not None else 64, cam, mp, op)
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
        if self.lsd is not None:  # key-lines: records from the detector, descriptors from LBD, end points as the rows the ingest takes
            M = max_kl
            self.kl = torch.zeros((2 * B, M, 6), dtype=torch.float32, device=dev)   # stvo_keyline records (24 bytes)
            self.kl_xy = torch.zeros((2 * B, M, 4), dtype=torch.float32, device=dev)
            self.ldesc = torch.zeros((2 * B, M, 32), dtype=torch.uint8, device=dev)
            self.nl = torch.zeros((2 * B,), dtype=torch.int32, device=dev)
            ff.stride_kl = M
            ff.n_kl_l = C.c_void_p(self.nl.data_ptr())
            ff.n_kl_r = C.c_void_p(self.nl.data_ptr() + 4 * B)
            ff.kl_l = C.c_void_p(self.kl_xy.data_ptr())
            ff.kl_r = C.c_void_p(self.kl_xy.data_ptr() + 16 * B * M)
            ff.ldesc_l = C.c_void_p(self.ldesc.data_ptr())
            ff.ldesc_r = C.c_void_p(self.ldesc.data_ptr() + 32 * B * M)
        self.ff = ff  # (without lsd every line pointer stays NULL: no key-lines; oct_ll NULL: one octave)
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
        if self.lsd is not None:
            ip = img_ptr if img_ptr is not None else self.img.data_ptr()
            self.lsd.detect_dev(ip, self.kl.data_ptr(), None, self.nl.data_ptr())
            self.lbd.compute_dev(ip, self.kl.data_ptr(), self.nl.data_ptr(), self.ldesc.data_ptr())
            self.ctx._chk(self.ctx.lib.stvo_keylines_xy_dev(self.ctx.h, 2 * self.B, self.M, self.kl.data_ptr(), self.nl.data_ptr(), self.kl_xy.data_ptr()))
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
        if self.lsd is not None:
            self.lsd.close()
            self.lbd.close()
