"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's pose-error metric (MPJAE).

Only ``tests/`` may import this module; the product computes the metric on the GPU
(``keypoints2body_amd/evaluation.py`` -> ``k2b_angular_error_deg``).  Pinned by
``tests/golden/mpjae.npz``, generated with the real reference (``oracle/gen_golden_io_eval.py``).

Follows reference ``keypoints2body/cli/eval.py``:
  rotvec_to_rotmat          :88-128   (float32; Taylor branch for |theta| <= 1e-8)
  compute_angular_error_deg :131-140  (trace of R_pred . R_gt elementwise, clip to [-1+1e-6, 1-1e-6], arccos)
  evaluate_pose_pair        :143-160  (common prefix of frames / whole rotations, float64 sum)
"""
from __future__ import annotations

import numpy as np


def rotvec_to_rotmat(rotvec: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    v = np.asarray(rotvec, dtype=np.float32)
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    t2 = x * x + y * y + z * z
    t = np.sqrt(t2)
    small = t <= eps
    ts = np.where(small, np.float32(1.0), t).astype(np.float32)
    a = np.sin(ts) / ts
    b = (np.float32(1.0) - np.cos(ts)) / (ts * ts)
    if small.any():
        s2 = t2[small]
        a[small] = 1.0 - s2 / 6.0 + (s2 * s2) / 120.0
        b[small] = 0.5 - s2 / 24.0 + (s2 * s2) / 720.0
    R = np.empty(v.shape[:-1] + (3, 3), dtype=np.float32)
    R[..., 0, 0] = 1.0 - b * (y * y + z * z)
    R[..., 0, 1] = b * (x * y) - a * z
    R[..., 0, 2] = b * (x * z) + a * y
    R[..., 1, 0] = b * (x * y) + a * z
    R[..., 1, 1] = 1.0 - b * (x * x + z * z)
    R[..., 1, 2] = b * (y * z) - a * x
    R[..., 2, 0] = b * (x * z) - a * y
    R[..., 2, 1] = b * (y * z) + a * x
    R[..., 2, 2] = 1.0 - b * (x * x + y * y)
    return R


def compute_angular_error_deg(pred_rotvec: np.ndarray, gt_rotvec: np.ndarray) -> np.ndarray:
    trace = np.sum(rotvec_to_rotmat(pred_rotvec) * rotvec_to_rotmat(gt_rotvec), axis=(-1, -2))
    c = np.clip((trace - 1.0) * 0.5, -1.0 + 1e-6, 1.0 - 1e-6)
    return np.degrees(np.arccos(c))


def evaluate_pose_pair(pred_pose: np.ndarray, gt_pose: np.ndarray):
    n = min(gt_pose.shape[0], pred_pose.shape[0])
    d = (min(gt_pose.shape[1], pred_pose.shape[1]) // 3) * 3
    ang = compute_angular_error_deg(pred_pose[:n, :d].reshape(n, d // 3, 3), gt_pose[:n, :d].reshape(n, d // 3, 3))
    total = float(np.sum(ang, dtype=np.float64))
    return total / ang.size, total, int(ang.size)
