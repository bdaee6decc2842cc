"""Frame-fitting coordinator and initialisation helpers.

Host plumbing that keeps the reference's engine interface
(reference ``keypoints2body/core/engine.py:23-262``): ``OptimizeEngine.fit_frame`` routes to the
configured estimator; the init builders create mean-pose / zero parameters and the root-aligned
initial translation (one joints-only LBS launch, row A10 of SURVEY.md §8a).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from ..models.body_model import as_body_model
from ..models.smpl_data import BodyModelFitResult, BodyModelParams, SMPLData, SMPLHData, SMPLXData
from .config import FrameOptimizeConfig, SequenceOptimizeConfig
from .estimators.factory import create_estimator
from .fitters.world_space import guess_init_transl_from_root


class OptimizeEngine:
    """Routes frame fitting to the estimator selected by the frame config."""

    def __init__(self, model, frame_config: FrameOptimizeConfig, device=None, model_type: str = "smpl",
                 pose_prior=None):
        self.model = as_body_model(model, device=device) if model is not None else None
        self.frame_config = frame_config
        self.device = self.model.device if self.model is not None else device
        self.estimator = create_estimator(model=self.model, frame_config=frame_config, device=self.device,
                                          model_type=model_type, pose_prior=pose_prior)

    def fit_frame(self, init_params: BodyModelParams, j3d: torch.Tensor, conf_3d: Optional[torch.Tensor],
                  seq_ind: int, target_model_indices: Optional[torch.Tensor] = None) -> BodyModelFitResult:
        return self.estimator.fit_frame(init_params=init_params, j3d=j3d, conf_3d=conf_3d, seq_ind=seq_ind,
                                        target_model_indices=target_model_indices)


def load_mean_pose_shape(mean_file: str, device) -> tuple[torch.Tensor, torch.Tensor]:
    """Mean pose (1,72) and mean shape (1,10) from ``neutral_smpl_mean_params.h5`` (datasets
    ``pose`` / ``shape``, reference ``core/engine.py:71-86``) or from an ``.npz`` twin with the
    same two arrays (``h5py`` is an optional dependency here)."""
    npz = os.path.splitext(mean_file)[0] + ".npz"
    if mean_file.endswith(".npz") or (not os.path.exists(mean_file) and os.path.exists(npz)):
        with np.load(mean_file if mean_file.endswith(".npz") else npz) as z:
            pose, shape = z["pose"], z["shape"]
    else:
        if not os.path.exists(mean_file):
            raise FileNotFoundError(f"mean parameter file not found: {mean_file}")
        try:
            import h5py
        except ImportError as e:  # pragma: no cover - depends on the environment
            raise ImportError(f"reading {mean_file} needs h5py; alternatively provide {npz} with arrays "
                              "'pose' (72,) and 'shape' (10,)") from e
        with h5py.File(mean_file, "r") as f:
            pose, shape = f["pose"][:], f["shape"][:]
    to = lambda a: torch.as_tensor(np.asarray(a)).unsqueeze(0).float().to(device)
    return to(pose), to(shape)


def default_init_params(mean_pose, mean_shape, joints_frame, model, joints_category: str,
                        coordinate_mode: str) -> SMPLData:
    """Mean pose / shape plus, in world mode, the root-aligned translation
    (reference ``core/engine.py:89-128``)."""
    pose = mean_pose.clone().detach()
    betas = mean_shape.clone().detach()
    transl = None
    if coordinate_mode == "world":
        if joints_category == "GENERIC":
            m = as_body_model(model)
            out = m(global_orient=pose[:, :3], body_pose=pose[:, 3:], betas=betas, return_verts=False)
            target = torch.as_tensor(joints_frame, dtype=torch.float32).to(m.device)
            transl = (target[:, 0, :] - out.joints[:, 0, :]).detach()
        else:
            transl = guess_init_transl_from_root(model, pose, betas, joints_frame, joints_category=joints_category)
    return SMPLData(betas=betas, global_orient=pose[:, :3], body_pose=pose[:, 3:], transl=transl)


def upgrade_smpl_family_init_params(base: SMPLData, model_type: str, model, device) -> BodyModelParams:
    """SMPL init -> SMPL-H / SMPL-X init with zero hands / face (reference ``core/engine.py:170-214``)."""
    if model_type == "smpl":
        return base
    n = base.body_pose.shape[0]
    zeros = lambda c: torch.zeros((n, c), device=device)
    common = dict(betas=base.betas, global_orient=base.global_orient, body_pose=base.body_pose, transl=base.transl)
    hand = int(getattr(model, "NUM_HAND_JOINTS", 15)) * 3
    if model_type == "smplh":
        return SMPLHData(**common, left_hand_pose=zeros(hand), right_hand_pose=zeros(hand))
    if model_type == "smplx":
        expr = int(getattr(model, "num_expression_coeffs", 10))
        return SMPLXData(**common, left_hand_pose=zeros(hand), right_hand_pose=zeros(hand), expression=zeros(expr),
                         jaw_pose=zeros(3), leye_pose=zeros(3), reye_pose=zeros(3))
    raise ValueError(f"Unsupported SMPL-family model_type: {model_type}")


def optimize_shape_pass(model, seq_config: SequenceOptimizeConfig, init_mean_shape, init_mean_pose, data_tensor,
                        confidence_input, device):
    """Multi-frame shared-betas pre-pass (reference ``core/engine.py:217-262`` ->
    ``core/shape.py:10-115``).  Not built on the HIP engine yet; note that the
    reference's own Adam branch of this pass raises (``shape.py:10,110-113``), so Adam runs of the
    reference need ``use_shape_optimization=False`` as well."""
    if not seq_config.use_shape_optimization:
        return init_mean_shape
    raise NotImplementedError(
        "use_shape_optimization=True: the shared-shape pre-pass is not built on the HIP engine yet; "
        "set SequenceOptimizeConfig(use_shape_optimization=False)")
