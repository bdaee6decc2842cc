"""Motion files in, renderer zip out: the data formats either side of the fitting path.

Behavioural mirror of the reference's ``keypoints2body/io/motion.py`` (host code only; nothing here
touches the GPU):

* ``load_motion_data`` (reference ``io/motion.py:16-56``): ``.npy`` array, ``.npz`` with a ``joints``
  entry, or the motion-capture ``.csv`` export (five header rows, two leading index/time columns, then
  x,y,z per joint); the joints then pass through ``adapt_layout`` and a warning reports a change of the
  joint count.
* ``write_smplx_zip`` (``io/motion.py:59-107``): one ``frame_%06d/person_%02d.npz`` member per frame with
  the SMPL-X field set a renderer expects (21 body joints, zero hands / jaw / expression).
* ``write_smplx_zip_from_smpl_data`` (``io/motion.py:110-138``).
"""
from __future__ import annotations

import csv
import io
import warnings
import zipfile
from pathlib import Path
from typing import Optional, Tuple

import numpy as np
import torch

from ..core.joints.adapters import adapt_layout
from ..models.smpl_data import SMPLData

_CSV_HEADER_ROWS = 5     # rows above the data in the capture export
_CSV_LEAD_COLS = 2       # frame index and time stamp


def _read_capture_csv(path: Path) -> np.ndarray:
    frames = []
    with open(path, "r", newline="") as fh:
        rows = csv.reader(fh)
        for _ in range(_CSV_HEADER_ROWS):
            next(rows)
        for row in rows:
            vals = [float(v) for v in row[_CSV_LEAD_COLS:]]
            n = len(vals) // 3          # a trailing incomplete triple is dropped, as zip() does in the reference
            frames.append(np.asarray(vals[:3 * n], dtype=np.float64).reshape(n, 3))
    return np.array(frames)


def load_motion_data(path: Path, layout: Optional[str] = None) -> Tuple[np.ndarray, str, int]:
    """Read 3D joints from ``path`` and bring them to a canonical layout.

    Returns ``(joints (T,K,3), layout name, K)``.  ``ValueError`` for an unknown extension or an ``.npz``
    without a ``joints`` entry (reference ``io/motion.py:43-47``).
    """
    path = Path(path)
    ext = path.suffix.lower()
    if ext == ".npy":
        joints = np.load(path)
    elif ext == ".csv":
        joints = _read_capture_csv(path)
    elif ext == ".npz":
        with np.load(path) as data:
            if "joints" not in data:
                raise ValueError(f"Unsupported .npz format: found keys {list(data.keys())}")
            joints = data["joints"]
    else:
        raise ValueError(f"Unsupported 3D joints file format: {ext}")

    k_in = joints.shape[1]
    joints, out_layout = adapt_layout(joints, layout)
    if joints.shape[1] != k_in:
        warnings.warn(f"Converted input joints from {k_in} to {joints.shape[1]} for layout {out_layout}.")
    return joints, out_layout, joints.shape[1]


def write_smplx_zip(output_dir: Path, poses: np.ndarray, betas: np.ndarray, transl: np.ndarray,
                    zip_name: str = "smpl_params.zip", person_idx: int = 0) -> Path:
    """Write (T,72) SMPL poses, betas ((10,) or (T,10)) and (T,3) translations as a renderer zip."""
    zip_path = Path(output_dir).expanduser() / zip_name
    if poses.ndim != 2 or poses.shape[1] != 72:
        raise ValueError(f"Expected poses shape (T,72); got {poses.shape}")
    T = poses.shape[0]
    if betas.ndim == 1:
        betas = np.broadcast_to(betas[None, :], (T, betas.shape[0]))
    if betas.shape[0] != T:
        raise ValueError(f"Expected betas shape (T,10); got {betas.shape}")
    if transl.shape[0] != T:
        raise ValueError(f"Expected transl shape (T,3); got {transl.shape}")

    f32 = np.float32
    constant = dict(expression=np.zeros(10, f32), left_hand_pose=np.zeros(45, f32), right_hand_pose=np.zeros(45, f32),
                    jaw_pose=np.zeros(3, f32))
    with zipfile.ZipFile(zip_path, "w", compression=zipfile.ZIP_DEFLATED) as zf:
        for t in range(T):
            # key order follows the reference's dict (it is the member order inside each .npz)
            fields = dict(betas=np.asarray(betas[t], f32), expression=constant["expression"],
                          global_orient=np.asarray(poses[t, :3], f32), body_pose=np.asarray(poses[t, 3:66], f32),
                          left_hand_pose=constant["left_hand_pose"], right_hand_pose=constant["right_hand_pose"],
                          jaw_pose=constant["jaw_pose"], transl=np.asarray(transl[t], f32))
            buf = io.BytesIO()
            np.savez(buf, **fields)
            zf.writestr(f"frame_{t:06d}/person_{person_idx:02d}.npz", buf.getvalue())
    return zip_path


def write_smplx_zip_from_smpl_data(output_dir: Path, smpl_data: SMPLData, zip_name: str = "smpl_params.zip",
                                   person_idx: int = 0) -> Path:
    """``write_smplx_zip`` for one ``SMPLData`` sequence (device tensors are brought to the host)."""
    if smpl_data.transl is None:
        raise ValueError("smpl_data.transl is required to export SMPL-X zip")
    host = lambda x: x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x
    return write_smplx_zip(output_dir=output_dir, poses=host(smpl_data.pose), betas=host(smpl_data.betas),
                           transl=host(smpl_data.transl), zip_name=zip_name, person_idx=person_idx)
