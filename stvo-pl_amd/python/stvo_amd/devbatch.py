"""Device-resident batches for the throughput path (torch is used only to allocate HBM / streams)."""
import numpy as np

from .capi import TrackBatchDev
from .ctypes_types import POSE_RESULT_DTYPE


class TrackBatch:
    """B frame pairs laid out as include/stvo_hip.h:stvo_track_batch_dev expects, on `device`."""

    def __init__(self, frames, max_pts, max_lines=0, device="cuda:0"):
        import torch
        self.B = B = len(frames)
        self.max_pts, self.max_lines = max_pts, max_lines
        self.frames = frames
        h = {}
        h["n_prev_pts"] = np.array([len(f["prev_P"]) for f in frames], np.int32)
        h["n_curr_pts"] = np.array([len(f["curr_pl"]) for f in frames], np.int32)
        h["prev_pdesc"] = np.zeros((B, max_pts, 32), np.uint8)
        h["curr_pdesc"] = np.zeros((B, max_pts, 32), np.uint8)
        h["prev_P"] = np.ones((B, max_pts, 3))
        h["prev_sigma2p"] = np.ones((B, max_pts))
        h["curr_pl"] = np.zeros((B, max_pts, 2))
        for b, f in enumerate(frames):
            n1, n2 = h["n_prev_pts"][b], h["n_curr_pts"][b]
            h["prev_pdesc"][b, :n1] = f["prev_desc"]
            h["curr_pdesc"][b, :n2] = f["curr_desc"]
            h["prev_P"][b, :n1] = f["prev_P"]
            h["prev_sigma2p"][b, :n1] = f["prev_sigma2"]
            h["curr_pl"][b, :n2] = f["curr_pl"]
        if max_lines > 0:
            h["n_prev_lines"] = np.array([len(f["prev_sP"]) for f in frames], np.int32)
            h["n_curr_lines"] = np.array([len(f["curr_le"]) for f in frames], np.int32)
            h["prev_ldesc"] = np.zeros((B, max_lines, 32), np.uint8)
            h["curr_ldesc"] = np.zeros((B, max_lines, 32), np.uint8)
            for k, w in (("prev_sP", 3), ("prev_eP", 3), ("prev_spl", 2), ("prev_epl", 2), ("curr_le", 3)):
                h[k] = np.ones((B, max_lines, w))
            h["prev_sigma2l"] = np.ones((B, max_lines))
            for b, f in enumerate(frames):
                n1, n2 = h["n_prev_lines"][b], h["n_curr_lines"][b]
                h["prev_ldesc"][b, :n1] = f["prev_ldesc"]
                h["curr_ldesc"][b, :n2] = f["curr_ldesc"]
                for k in ("prev_sP", "prev_eP", "prev_spl", "prev_epl"):
                    h[k][b, :n1] = f[k]
                h["prev_sigma2l"][b, :n1] = f["prev_sigma2l"]
                h["curr_le"][b, :n2] = f["curr_le"]
        self.host = h
        self.dev = {k: torch.from_numpy(v).to(device) for k, v in h.items()}
        self.dev["m12_pts"] = torch.full((B, max_pts), -7, dtype=torch.int32, device=device)
        self.dev["inlier_pts"] = torch.full((B, max_pts), -7, dtype=torch.int32, device=device)
        self.dev["m12_lines"] = torch.full((B, max(max_lines, 1)), -7, dtype=torch.int32, device=device)
        self.dev["inlier_lines"] = torch.full((B, max(max_lines, 1)), -7, dtype=torch.int32, device=device)
        self.dev["results"] = torch.zeros(B * POSE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=device)
        s = TrackBatchDev()
        s.B, s.max_pts, s.max_lines = B, max_pts, max_lines
        for name, _ in TrackBatchDev._fields_[4:]:
            t = self.dev.get(name)
            setattr(s, name, t.data_ptr() if t is not None else None)
        self.struct = s

    def algorithmic_bytes_match(self):
        """compulsory bytes of the brute-force matcher, 32 (N1+N2) + 4 N1 per frame pair."""
        n1 = self.host["n_prev_pts"].astype(np.int64)
        n2 = self.host["n_curr_pts"].astype(np.int64)
        return int((32 * (n1 + n2) + 4 * n1).sum())

    def pairs_match(self):
        n1 = self.host["n_prev_pts"].astype(np.int64)
        n2 = self.host["n_curr_pts"].astype(np.int64)
        return int((n1 * n2).sum())

    def results(self):
        raw = self.dev["results"].cpu().numpy()
        return np.frombuffer(raw.tobytes(), dtype=POSE_RESULT_DTYPE)

    def m12_pts(self):
        return self.dev["m12_pts"].cpu().numpy()

    def inlier_pts(self):
        return self.dev["inlier_pts"].cpu().numpy()

    def m12_lines(self):
        return self.dev["m12_lines"].cpu().numpy()

    def inlier_lines(self):
        return self.dev["inlier_lines"].cpu().numpy()
