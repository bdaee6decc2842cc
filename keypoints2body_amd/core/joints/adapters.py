"""Input adapters: bring user joints into the canonical tensors the fitters take.

Host-side index shuffles only (no kernels): same observable behaviour as the reference's
adapter layer (reference ``keypoints2body/core/joints/adapters.py:35-380``):

* a frame ``(K,3|4)`` becomes ``j3d (1,K,3)`` + ``conf (K,)``; a sequence ``(T,K,3|4)``
  becomes ``(T,K,3)`` + ``(T,K)``; a 4th channel is the confidence, otherwise ones;
* a layout label (or the joint count) selects a mapping into the AMASS-22 / SMPL-24 order,
  optionally with the OpenSim -> SMPL axis change;
* dict input (``body`` / ``left_hand`` / ``right_hand`` / ``face`` blocks) is concatenated and
  comes with explicit model joint indices (layout ``"GENERIC"``).
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import numpy as np
import torch

from ..constants import (AMASS_SMPL_IDX, SMPL_IDX, SMPLX_FACE_IDX_START, SMPLX_LEFT_HAND_IDX,
                         SMPLX_RIGHT_HAND_IDX)


class JointLayoutAdapter(NamedTuple):
    """One supported input layout: joint count, canonical output layout, optional re-indexing
    into that layout and whether coordinates are OpenSim-style (x fwd, y up, z right)."""

    name: str
    expected_joints: int
    out_layout: str
    mapping: Optional[tuple] = None
    rotate_osim_to_smpl: bool = False


# index of the source joint that supplies each of the 22 AMASS joints
_MANNY25 = (0, 21, 17, 1, 22, 18, 3, 23, 19, 5, 24, 20, 6, 13, 9, 8, 14, 10, 15, 11, 16, 12)
_SPINETRACK37 = (0, 31, 25, 1, 32, 26, 3, 33, 27, 5, 34, 28, 6, 21, 17, 8, 22, 18, 23, 19, 24, 20)

ADAPTERS = {
    a.name: a
    for a in (
        JointLayoutAdapter("SMPL24", 24, "SMPL24"),
        JointLayoutAdapter("AMASS", 22, "AMASS"),
        JointLayoutAdapter("Manny25", 25, "AMASS", _MANNY25, True),
        JointLayoutAdapter("Halpe26", 26, "AMASS", tuple(range(22)), False),
        JointLayoutAdapter("SpineTrack37", 37, "AMASS", _SPINETRACK37, True),
    )
}

# OpenSim (x, y, z) -> SMPL (z, y, -x)
_OSIM_TO_SMPL = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]])


def resolve_adapter(joint_count: int, layout: Optional[str]) -> JointLayoutAdapter:
    """Adapter for an explicit layout label, else the one whose joint count matches."""
    if layout is None:
        for ad in ADAPTERS.values():
            if ad.expected_joints == joint_count:
                return ad
        raise ValueError(f"Unsupported number of joints: {joint_count}")
    if layout not in ADAPTERS:
        raise ValueError(f"Unsupported layout: {layout}")
    ad = ADAPTERS[layout]
    if ad.expected_joints != joint_count:
        raise ValueError(f"Layout {layout} expects {ad.expected_joints} joints, got {joint_count}")
    return ad


def _remap(arr: np.ndarray, ad: JointLayoutAdapter) -> np.ndarray:
    return arr if ad.mapping is None else arr[:, list(ad.mapping)]


def adapt_layout(joints_seq: np.ndarray, layout: Optional[str]):
    """(T,K,3) in a known layout -> (T,K',3) in the canonical layout, plus its label."""
    ad = resolve_adapter(joints_seq.shape[1], layout)
    pts = _remap(joints_seq, ad)
    if ad.rotate_osim_to_smpl:
        pts = pts @ _OSIM_TO_SMPL.astype(pts.dtype).T
    return pts, ad.out_layout


def adapt_layout_and_conf(joints_seq: np.ndarray, conf_seq: np.ndarray, layout: Optional[str]):
    """As ``adapt_layout`` and the same re-indexing applied to (T,K) confidences."""
    ad = resolve_adapter(joints_seq.shape[1], layout)
    pts, out_layout = adapt_layout(joints_seq, layout)
    return pts, _remap(conf_seq, ad), out_layout


def _as_float_tensor(x, what: str) -> torch.Tensor:
    if isinstance(x, np.ndarray):
        return torch.as_tensor(x, dtype=torch.float32)
    if isinstance(x, torch.Tensor):
        return x.float()
    raise ValueError(f"{what} must be numpy array or torch tensor")


def normalize_joints_frame(joints):
    """(K,3) or (K,4) -> j3d (1,K,3), conf (K,)."""
    jt = _as_float_tensor(joints, "joints")
    if jt.ndim != 2 or jt.shape[1] not in (3, 4):
        raise ValueError(f"Expected joints shape (K,3) or (K,4), got {tuple(jt.shape)}")
    conf = jt[:, 3].clone() if jt.shape[1] == 4 else torch.ones(jt.shape[0], dtype=jt.dtype, device=jt.device)
    return jt[:, :3].unsqueeze(0), conf


def normalize_joints_sequence(joints_seq):
    """(T,K,3) or (T,K,4) -> xyz (T,K,3), conf (T,K)."""
    jt = _as_float_tensor(joints_seq, "joints_seq")
    if jt.ndim != 3 or jt.shape[2] not in (3, 4):
        raise ValueError(f"Expected joints_seq shape (T,K,3) or (T,K,4), got {tuple(jt.shape)}")
    if jt.shape[2] == 4:
        return jt[:, :, :3], jt[:, :, 3]
    return jt, torch.ones(jt.shape[:2], dtype=jt.dtype, device=jt.device)


# block name -> (bodies it is allowed with, required joint count or None, model joint indices)
_BLOCKS = (
    ("body", None, None),
    ("left_hand", {"smplh", "smplx"}, list(SMPLX_LEFT_HAND_IDX)),
    ("right_hand", {"smplh", "smplx"}, list(SMPLX_RIGHT_HAND_IDX)),
    ("face", {"smplx"}, None),
)


def _block_indices(name: str, count: int, body_model: str) -> torch.Tensor:
    if name == "body":
        if count == 24:
            return torch.tensor(list(SMPL_IDX), dtype=torch.long)
        if count == 22:
            return torch.tensor(list(AMASS_SMPL_IDX), dtype=torch.long)
        raise ValueError("body block must have 22 or 24 joints")
    if name in ("left_hand", "right_hand"):
        if body_model not in {"smplh", "smplx"}:
            raise ValueError(f"{name} block requires body_model='smplh' or 'smplx'")
        if count != 21:
            raise ValueError(f"{name} block must have 21 joints")
        return torch.tensor(list(SMPLX_LEFT_HAND_IDX if name == "left_hand" else SMPLX_RIGHT_HAND_IDX),
                            dtype=torch.long)
    if body_model != "smplx":
        raise ValueError("face block requires body_model='smplx'")
    return torch.arange(SMPLX_FACE_IDX_START, SMPLX_FACE_IDX_START + count, dtype=torch.long)


def _normalize_blocks(blocks: dict, body_model: str, sequence: bool):
    pts, confs, idx, frames = [], [], [], None
    for name, _, _ in _BLOCKS:
        if name not in blocks:
            continue
        p, c = (normalize_joints_sequence if sequence else normalize_joints_frame)(blocks[name])
        if sequence:
            if frames is None:
                frames = p.shape[0]
            elif p.shape[0] != frames:
                raise ValueError("all dict sequence blocks must share same T")
        idx.append(_block_indices(name, p.shape[1], body_model))
        pts.append(p)
        confs.append(c)
    if not pts:
        raise ValueError("dict input must provide at least one of: body, left_hand, right_hand, face")
    return torch.cat(pts, dim=1), torch.cat(confs, dim=1 if sequence else 0), torch.cat(idx, dim=0), "GENERIC"


def normalize_frame_observations(joints, *, layout: Optional[str], body_model: str):
    """One frame (array or dict of blocks) -> (j3d (1,K,3), conf (K,), model indices | None, label)."""
    if isinstance(joints, dict):
        return _normalize_blocks(joints, body_model, sequence=False)
    j3d, conf = normalize_joints_frame(joints)
    return j3d, conf, None, "AUTO"


def normalize_sequence_observations(joints_seq, *, layout: Optional[str], body_model: str):
    """A sequence (array or dict of blocks) -> (xyz (T,K,3), conf (T,K), model indices | None, label)."""
    if isinstance(joints_seq, dict):
        return _normalize_blocks(joints_seq, body_model, sequence=True)
    xyz, conf = normalize_joints_sequence(joints_seq)
    return xyz, conf, None, "AUTO"
