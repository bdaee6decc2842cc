"""``optimize_params_frame``: fit one frame of 3D joints (reference
``keypoints2body/api/frame.py:34-219``), executed on the HIP engine."""
from __future__ import annotations

from typing import Optional

import torch

from ..core.config import FrameOptimizeConfig, ModelType, frame_config_from
from ..core.engine import (OptimizeEngine, default_init_params, load_mean_pose_shape,
                           upgrade_smpl_family_init_params)
from ..core.joints.adapters import normalize_frame_observations
from ..models.smpl_data import BodyModelFitResult, BodyModelParams, SMPLData, SMPLHData, SMPLXData
from . import common


def _with_root_aligned_transl(prev: BodyModelParams, j3d, model, frame_cfg, device) -> BodyModelParams:
    """Warm-start parameters that lack ``transl`` get the root-aligned one
    (reference ``api/frame.py:164-211``); hand / face fields are carried over."""
    pose = torch.as_tensor(prev.pose, dtype=torch.float32, device=device)
    betas = torch.as_tensor(prev.betas, dtype=torch.float32, device=device)
    transl = default_init_params(pose, betas, j3d, model, frame_cfg.joints_category, frame_cfg.coordinate_mode).transl
    base = dict(betas=betas, global_orient=pose[:, :3], body_pose=pose[:, 3:], transl=transl,
                metadata=dict(getattr(prev, "metadata", {})))
    if isinstance(prev, SMPLXData):
        return SMPLXData(**base, left_hand_pose=prev.left_hand_pose, right_hand_pose=prev.right_hand_pose,
                         expression=prev.expression, jaw_pose=prev.jaw_pose, leye_pose=prev.leye_pose,
                         reye_pose=prev.reye_pose)
    if isinstance(prev, SMPLHData):
        return SMPLHData(**base, left_hand_pose=prev.left_hand_pose, right_hand_pose=prev.right_hand_pose)
    return SMPLData(**base)


def optimize_params_frame(joints, *, prev_params: Optional[BodyModelParams] = None, body_model: ModelType = "smpl",
                          joint_layout: Optional[str] = None, model=None,
                          config: Optional[FrameOptimizeConfig | dict] = None, device=None,
                          pose_prior=None, mean_params: Optional[tuple] = None) -> BodyModelFitResult:
    """Optimise body parameters for a single frame of 3D joints.

    Same arguments as the reference plus two optional conveniences for setups without the
    licensed asset files: ``pose_prior`` (a ``keypoints2body_amd.prior.MaxMixturePrior``) and
    ``mean_params`` (``(mean_pose (1,72), mean_shape (1,NB))`` tensors) replace the files the
    reference reads from ``./data/models``.
    """
    device = common.resolve_device(device)
    frame_cfg = frame_config_from(config)
    common.check_request(frame_cfg, body_model)
    j3d, conf_3d, model_indices, in_layout = normalize_frame_observations(joints, layout=joint_layout,
                                                                         body_model=body_model)
    if body_model in common.SMPL_FAMILY and in_layout != "GENERIC":
        j3d, conf2d, model_indices = common.canonicalize(j3d, conf_3d[None], None, in_layout, joint_layout,
                                                         body_model, frame_cfg, device)
        conf_3d = conf2d[0]
    else:
        j3d, conf_3d, model_indices = common.canonicalize(j3d, conf_3d, model_indices, in_layout, joint_layout,
                                                          body_model, frame_cfg, device)
    model = common.obtain_model(model, body_model, device)
    engine = OptimizeEngine(model=model, frame_config=frame_cfg, device=device, model_type=body_model,
                            pose_prior=pose_prior)

    if prev_params is None:
        mean_pose, mean_shape = mean_params if mean_params is not None else load_mean_pose_shape(
            common.DEFAULT_MEAN_FILE, device)
        base = default_init_params(mean_pose.to(device), mean_shape.to(device), j3d, model,
                                   joints_category=frame_cfg.joints_category,
                                   coordinate_mode=frame_cfg.coordinate_mode)
        init_params = upgrade_smpl_family_init_params(base, model_type=body_model, model=model, device=device)
    else:
        common.check_param_type(prev_params, body_model, "prev_params")
        init_params = prev_params.to(device)
        if frame_cfg.coordinate_mode == "world" and init_params.transl is None:
            init_params = _with_root_aligned_transl(init_params, j3d, model, frame_cfg, device)

    return engine.fit_frame(init_params=init_params, j3d=j3d, conf_3d=conf_3d, seq_ind=0,
                            target_model_indices=model_indices)
