"""Pose-error evaluation (MPJAE) on the GPU: the consumer side of the fitting path.

Mirrors the library part of the reference's ``keypoints2body/cli/eval.py`` (the argument parser and the
progress-bar loop around it are CLI and out of scope):

* ``discover_amass_npz_files`` (``cli/eval.py:54-59``), ``load_amass_sequence`` (``:62-84``),
  ``save_prediction_pose`` (``:199-210``): host I/O, same behaviour;
* ``compute_angular_error_deg`` (``:131-140``) and ``evaluate_pose_pair`` (``:143-160``): the rotations go to
  the device once and ``k2b_angular_error_deg`` (csrc/k2b_metrics.hip) evaluates every pair in one launch; the
  sum is taken in float64 on the device, as the reference sums in float64 on the host.

Like the rest of the package there is no CPU path: without a HIP device these functions raise.
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Tuple

import numpy as np
import torch

from . import native


def discover_amass_npz_files(root: Path) -> List[Path]:
    """A single file, or every ``*.npz`` under ``root`` in sorted order."""
    root = Path(root).expanduser().resolve()
    if root.is_file():
        return [root]
    return sorted(p for p in root.rglob("*.npz") if p.is_file())


def load_amass_sequence(npz_path: Path) -> Tuple[np.ndarray, np.ndarray]:
    """``(joints (T,22,3), gt_pose (T,72))`` of one evaluation sequence, cut to the common length.

    ``KeyError`` if ``joints`` / ``global_orient`` / ``body_pose`` is missing, ``ValueError`` for an empty
    sequence (reference ``cli/eval.py:65-84``)."""
    with np.load(npz_path) as data:
        missing = [k for k in ("joints", "global_orient", "body_pose") if k not in data]
        if missing:
            raise KeyError(f"Missing keys {missing} in {npz_path}")
        joints = np.asarray(data["joints"], dtype=np.float32)[:, :22, :]
        go = np.asarray(data["global_orient"], dtype=np.float32)
        bp = np.asarray(data["body_pose"], dtype=np.float32)
    go = go[None, :] if go.ndim == 1 else go
    bp = bp[None, :] if bp.ndim == 1 else bp
    gt_pose = np.concatenate([go, bp], axis=1).astype(np.float32)
    T = min(joints.shape[0], gt_pose.shape[0])
    if T == 0:
        raise ValueError(f"Empty sequence in {npz_path}")
    return joints[:T], gt_pose[:T]


def _device_f32(x, device) -> torch.Tensor:
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, dtype=np.float32))
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def compute_angular_error_deg(pred_rotvec, gt_rotvec, device=None) -> torch.Tensor:
    """Angular error in degrees between rotations given as axis-angle vectors ``(..., 3)``.
    Returns a device tensor of shape ``(...)``."""
    dev = native.require_device(device if device is not None else (pred_rotvec.device if isinstance(pred_rotvec, torch.Tensor) and pred_rotvec.is_cuda else None))
    return native.angular_error_deg(_device_f32(pred_rotvec, dev), _device_f32(gt_rotvec, dev))


def evaluate_pose_pair(pred_pose, gt_pose, device=None) -> Tuple[float, float, int]:
    """MPJAE of predicted vs ground-truth poses ``(T, 3k)``: ``(mean, sum, count)`` over the common frames
    and the common whole rotations (reference ``cli/eval.py:143-160``)."""
    n = min(gt_pose.shape[0], pred_pose.shape[0])
    d = (min(gt_pose.shape[1], pred_pose.shape[1]) // 3) * 3
    dev = native.require_device(device)
    pred = _device_f32(pred_pose, dev)[:n, :d].reshape(n, d // 3, 3).contiguous()
    gt = _device_f32(gt_pose, dev)[:n, :d].reshape(n, d // 3, 3).contiguous()
    ang = native.angular_error_deg(pred, gt)
    total = float(ang.double().sum())
    count = int(ang.numel())
    return total / count, total, count


def save_prediction_pose(pred_pose: np.ndarray, src_path: Path, dataset_root: Path, save_root: Path) -> None:
    """``save_root / <path of src relative to the dataset root>`` as a compressed npz with key ``pose``."""
    save_root = Path(save_root)
    save_root.mkdir(parents=True, exist_ok=True)
    try:
        out_path = save_root / Path(src_path).resolve().relative_to(Path(dataset_root).resolve())
    except ValueError:
        out_path = save_root / Path(src_path).name
    out_path.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(out_path, pose=pred_pose)
