"""Configuration dataclasses of the fitting path.

Field names, defaults and meaning follow the reference's only "flag system"
(reference ``keypoints2body/core/config.py:7-59``) so that a config object or dict
written for the reference configures this engine unchanged.  Loss weights that the
reference keeps as function defaults (``core/losses.py:33-38``: sigma 100, pose prior
4.78*1.5, angle prior 15.2, shape prior 5.0) are *not* config fields there either; the
HIP kernel receives them through ``k2b_fit_config`` (include/k2b.h).
"""
from __future__ import annotations

import dataclasses as _dc
from pathlib import Path
from typing import Literal, Optional

ModelType = Literal["smpl", "smplh", "smplx", "mano", "flame"]

_ESTIMATORS = ("optimization", "learned", "ikgat")
_INPUTS = ("joints3d", "joints2d", "multiview_joints2d")
_MODES = ("camera", "world")
_CATEGORIES = ("SMPL24", "AMASS", "GENERIC")


@_dc.dataclass
class BodyModelConfig:
    """Where and how to load a body model."""

    model_type: ModelType = "smpl"
    model_family: str = "smpl_family"
    gender: str = "neutral"
    ext: Optional[str] = None
    batch_size: int = 1
    model_dir: Path = Path("./data/models/")


@_dc.dataclass
class FrameOptimizeConfig:
    """Per-frame estimator settings."""

    estimator_type: Literal["optimization", "learned", "ikgat"] = "optimization"
    input_type: Literal["joints3d", "joints2d", "multiview_joints2d"] = "joints3d"
    coordinate_mode: Literal["camera", "world"] = "world"
    use_lbfgs: bool = True
    step_size: float = 1e-2
    num_iters: int = 100
    num_iters_first: int = 30
    num_iters_followup: int = 10
    joint_loss_weight: float = 600.0
    pose_preserve_weight: float = 5.0
    freeze_betas: bool = False
    shape_prior_weight: float = 5.0
    pose_prior_num_gaussians: int = 8
    joints_category: Literal["SMPL24", "AMASS", "GENERIC"] = "AMASS"
    ikgat_model_dir: Path = Path("./data/estimators")
    ikgat_model_format: str = "manny"
    ikgat_model_type: str = "pos_to_rot6"
    ikgat_parent_ids: Optional[list[int]] = None
    ikgat_hidden_dim: int = 128
    ikgat_num_layers: int = 3
    ikgat_num_heads: int = 4

    def check(self) -> None:
        """Value checks the reference leaves to downstream failures."""
        if self.estimator_type not in _ESTIMATORS:
            raise ValueError(f"Unknown estimator_type: {self.estimator_type}")
        if self.input_type not in _INPUTS:
            raise ValueError(f"Unknown input_type: {self.input_type}")
        if self.coordinate_mode not in _MODES:
            raise ValueError(f"Unknown coordinate_mode: {self.coordinate_mode}")
        if self.joints_category not in _CATEGORIES:
            raise ValueError("No such joints category!")


@_dc.dataclass
class SequenceOptimizeConfig:
    """Sequence-level behaviour around the per-frame fit."""

    frame: FrameOptimizeConfig = _dc.field(default_factory=FrameOptimizeConfig)
    num_shape_iters: int = 40
    num_shape_frames: int = 50
    use_shape_optimization: bool = True
    use_previous_frame_init: bool = True
    fix_foot: bool = False
    limit_frames: Optional[int] = None


def frame_config_from(config) -> FrameOptimizeConfig:
    """dict | FrameOptimizeConfig | None -> FrameOptimizeConfig (reference ``api/frame.py:64-69``)."""
    if isinstance(config, dict):
        return FrameOptimizeConfig(**config)
    if isinstance(config, FrameOptimizeConfig):
        return config
    return FrameOptimizeConfig()


def sequence_config_from(config) -> SequenceOptimizeConfig:
    """dict | SequenceOptimizeConfig | None -> SequenceOptimizeConfig (reference ``api/sequence.py:69-83``)."""
    if isinstance(config, dict):
        known = {f.name for f in _dc.fields(SequenceOptimizeConfig)} - {"frame"}
        kwargs = {k: config[k] for k in known if k in config}
        return SequenceOptimizeConfig(frame=FrameOptimizeConfig(**config.get("frame", {})), **kwargs)
    if isinstance(config, SequenceOptimizeConfig):
        return config
    return SequenceOptimizeConfig()
