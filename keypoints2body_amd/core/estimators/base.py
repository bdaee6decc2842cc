"""The estimator plug-in seam (reference ``keypoints2body/core/estimators/base.py:10-20``)."""
from __future__ import annotations

from typing import Optional, Protocol

import torch

from ...models.smpl_data import BodyModelFitResult, BodyModelParams


class BodyEstimator(Protocol):
    """Anything that turns one frame of target joints into body-model parameters."""

    def fit_frame(self, init_params: BodyModelParams, j3d: torch.Tensor, conf_3d: Optional[torch.Tensor],
                  seq_ind: int, target_model_indices: Optional[torch.Tensor] = None) -> BodyModelFitResult: ...
