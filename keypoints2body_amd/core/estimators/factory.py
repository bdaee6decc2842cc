"""Estimator selection by ``FrameOptimizeConfig.estimator_type``
(reference ``keypoints2body/core/estimators/factory.py:20-44``)."""
from __future__ import annotations

from ..config import FrameOptimizeConfig
from .optimization import OptimizationEstimator


def create_estimator(model, frame_config: FrameOptimizeConfig, device=None, model_type: str = "smpl", pose_prior=None):
    kind = frame_config.estimator_type
    if kind == "optimization":
        return OptimizationEstimator(model=model, frame_config=frame_config, device=device, model_type=model_type,
                                     pose_prior=pose_prior)
    if kind == "learned":
        # same behaviour as the reference's LearnedEstimator placeholder (factory.py:10-17)
        raise NotImplementedError(
            "estimator_type='learned' is not implemented yet. "
            "Implement under keypoints2body.core.estimators and wire model loading/inference.")
    if kind == "ikgat":
        raise NotImplementedError(
            "estimator_type='ikgat': the learned GAT regressor (reference core/estimators/ikgat/) needs "
            "torch_geometric and trained weights and is outside the HIP engine's scope")
    raise ValueError(f"Unknown estimator_type: {kind}")
