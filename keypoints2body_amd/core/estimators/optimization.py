"""Optimisation-based estimator: routes ``fit_frame`` to the HIP world-space fitter.

Counterpart of the reference's ``OptimizationEstimator``
(reference ``keypoints2body/core/estimators/optimization.py:14-85``): it picks the fitter
from ``model_type`` / ``coordinate_mode`` and injects ``joint_loss_weight``,
``pose_preserve_weight`` and ``freeze_betas`` from the frame config on every call.
World- and camera-space SMPL-family fitters exist on the HIP engine; the MANO / FLAME branches of
the reference's dispatch raise ``NotImplementedError``.
"""
from __future__ import annotations

from typing import Optional

import torch

from ...models.smpl_data import BodyModelFitResult, BodyModelParams
from ..config import FrameOptimizeConfig
from ..fitters.camera_space import CameraSpaceFitter
from ..fitters.world_space import WorldSpaceFitter


class OptimizationEstimator:
    def __init__(self, model, frame_config: FrameOptimizeConfig, device=None, model_type: str = "smpl",
                 pose_prior=None):
        self.frame_config = frame_config
        if model_type in ("mano", "flame"):
            raise NotImplementedError(
                f"body_model='{model_type}': the MANO/FLAME fitters (reference core/fitters/misc_models.py) "
                "are outside the HIP engine's scope")
        if frame_config.coordinate_mode == "camera":
            self._fitter = CameraSpaceFitter(
                smpl_model=model, step_size=frame_config.step_size, num_iters=frame_config.num_iters,
                use_lbfgs=frame_config.use_lbfgs, joints_category=frame_config.joints_category, device=device,
                pose_prior_num_gaussians=frame_config.pose_prior_num_gaussians, pose_prior=pose_prior)
            return
        self._fitter = WorldSpaceFitter(
            smpl_model=model, step_size=frame_config.step_size, num_iters_first=frame_config.num_iters_first,
            num_iters_followup=frame_config.num_iters_followup, use_lbfgs=frame_config.use_lbfgs,
            joints_category=frame_config.joints_category, device=device,
            pose_prior_num_gaussians=frame_config.pose_prior_num_gaussians, pose_prior=pose_prior)

    @property
    def fitter(self):
        return self._fitter

    def _weights(self):
        c = self.frame_config
        return dict(joint_loss_weight=c.joint_loss_weight, pose_preserve_weight=c.pose_preserve_weight,
                    freeze_betas=c.freeze_betas)

    def fit_frame(self, init_params: BodyModelParams, j3d: torch.Tensor, conf_3d: Optional[torch.Tensor],
                  seq_ind: int, target_model_indices: Optional[torch.Tensor] = None) -> BodyModelFitResult:
        return self._fitter.fit_frame(init_params=init_params, j3d=j3d, conf_3d=conf_3d, seq_ind=seq_ind,
                                      target_model_indices=target_model_indices, **self._weights())

    def fit_batch(self, init_params, j3d, conf_3d, seq_ind, target_model_indices=None, per_frame_conf=False,
                  run_forward=True):
        """B independent frames in one launch (not part of the reference protocol)."""
        return self._fitter.fit_batch(init_params, j3d, conf_3d, seq_ind, target_model_indices,
                                      per_frame_conf=per_frame_conf, run_forward=run_forward, **self._weights())

    def fit_chain(self, init_params, j3d, conf_3d, target_model_indices=None, run_forward=True):
        """One sequence in warm-start mode as one launch (``WorldSpaceFitter.fit_chain``)."""
        return self._fitter.fit_chain(init_params, j3d, conf_3d, target_model_indices, run_forward=run_forward,
                                      **self._weights())
