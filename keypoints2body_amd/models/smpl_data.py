"""Typed parameter / result containers of the fitting path.

Same names, fields and behaviour as the reference's data contract
(reference ``keypoints2body/models/smpl_data.py:33-120``) so
that results of this engine can be handed to code written against the reference:
``.pose`` = cat(global_orient, body_pose), ``.to(device)``, ``.detach()``, a free-form
``metadata`` dict, numpy or torch fields.
"""
from __future__ import annotations

import dataclasses as _dc
from typing import Any, Optional, Union

import numpy as np
import torch

ArrayLike = Union[np.ndarray, torch.Tensor]

# fields every container treats as arrays when moving / detaching
_BASE_ARRAY_FIELDS = ("betas", "global_orient", "body_pose", "transl")


def _map_arrays(obj, names, fn):
    """dataclasses.replace with ``fn`` applied to the named, non-None array fields."""
    changes = {}
    for name in names:
        value = getattr(obj, name)
        if value is not None:
            changes[name] = fn(value)
    return _dc.replace(obj, **changes)


@_dc.dataclass
class BodyModelParams:
    """Shape, root orientation, body pose and optional translation of one or more frames."""

    betas: ArrayLike
    global_orient: ArrayLike
    body_pose: ArrayLike
    transl: Optional[ArrayLike] = None
    metadata: dict[str, Any] = _dc.field(default_factory=dict)

    @property
    def pose(self) -> ArrayLike:
        if isinstance(self.global_orient, torch.Tensor):
            return torch.cat((self.global_orient, self.body_pose), dim=-1)
        return np.concatenate((self.global_orient, self.body_pose), axis=-1)

    def validate(self) -> None:
        missing = [n for n in ("betas", "global_orient", "body_pose") if getattr(self, n) is None]
        if missing:
            raise ValueError("betas, global_orient, and body_pose are required")

    def to(self, device) -> "BodyModelParams":
        move = lambda x: x.to(device=device) if isinstance(x, torch.Tensor) and device is not None else x
        return _map_arrays(self, _BASE_ARRAY_FIELDS, move)

    def detach(self) -> "BodyModelParams":
        cut = lambda x: x.detach() if isinstance(x, torch.Tensor) else x
        return _map_arrays(self, _BASE_ARRAY_FIELDS, cut)


@_dc.dataclass
class SMPLData(BodyModelParams):
    """SMPL parameters."""


@_dc.dataclass
class SMPLHData(SMPLData):
    """SMPL-H parameters (adds the two hand poses)."""

    left_hand_pose: Optional[ArrayLike] = None
    right_hand_pose: Optional[ArrayLike] = None


@_dc.dataclass
class SMPLXData(SMPLHData):
    """SMPL-X parameters (adds expression, jaw and eye poses)."""

    expression: Optional[ArrayLike] = None
    jaw_pose: Optional[ArrayLike] = None
    leye_pose: Optional[ArrayLike] = None
    reye_pose: Optional[ArrayLike] = None


@_dc.dataclass
class MANOData(BodyModelParams):
    """MANO parameters."""

    hand_pose: Optional[ArrayLike] = None


@_dc.dataclass
class FLAMEData(BodyModelParams):
    """FLAME parameters."""

    expression: Optional[ArrayLike] = None
    jaw_pose: Optional[ArrayLike] = None
    neck_pose: Optional[ArrayLike] = None
    leye_pose: Optional[ArrayLike] = None
    reye_pose: Optional[ArrayLike] = None


@_dc.dataclass
class BodyModelFitResult:
    """What a fitter returns: parameters, final vertices / joints and the loss."""

    params: BodyModelParams
    vertices: torch.Tensor
    joints: torch.Tensor
    loss: Optional[torch.Tensor] = None
