"""Steps shared by the frame and sequence entry points."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ..core.config import BodyModelConfig, FrameOptimizeConfig
from ..core.joints.adapters import adapt_layout_and_conf
from ..models.body_model import as_body_model
from ..models.smpl_data import FLAMEData, MANOData, SMPLData, SMPLHData, SMPLXData
from ..native import require_device
from .model_factory import load_body_model

DEFAULT_MEAN_FILE = "./data/models/neutral_smpl_mean_params.h5"
OPTIMIZATION_BODY_MODELS = {"smpl", "smplh", "smplx", "mano", "flame"}
SMPL_FAMILY = {"smpl", "smplh", "smplx"}
PARAM_TYPES = {"smpl": SMPLData, "smplh": SMPLHData, "smplx": SMPLXData, "mano": MANOData, "flame": FLAMEData}


def resolve_device(device) -> torch.device:
    """The reference defaults to CPU (``api/frame.py:62``); this engine has no CPU path, so the
    default is the current HIP device and an explicit CPU device is an error."""
    return require_device(device)


def check_request(frame_cfg: FrameOptimizeConfig, body_model: str) -> None:
    if frame_cfg.input_type != "joints3d":
        raise NotImplementedError(
            f"input_type='{frame_cfg.input_type}' is not implemented in this release. "
            "Current APIs support only joints3d.")
    if body_model not in OPTIMIZATION_BODY_MODELS:
        raise ValueError(f"Unsupported body_model: {body_model}")
    if body_model in ("mano", "flame"):          # before any model file is opened or a launch is made
        raise NotImplementedError(
            f"body_model='{body_model}': the MANO/FLAME fitters (reference core/fitters/misc_models.py) are outside the "
            "HIP engine's scope; SMPL-family models (smpl, smplh, smplx) are built")


def canonicalize(xyz: torch.Tensor, conf: torch.Tensor, model_indices, in_layout: str, joint_layout: Optional[str],
                 body_model: str, frame_cfg: FrameOptimizeConfig, device):
    """Layout adaptation + joints_category bookkeeping (reference ``api/frame.py:82-104``,
    ``api/sequence.py:96-118``).  Like the reference this MUTATES ``frame_cfg.joints_category``."""
    if body_model in SMPL_FAMILY and in_layout != "GENERIC":
        pts, cf, out_layout = adapt_layout_and_conf(xyz.cpu().numpy(), conf.cpu().numpy(), joint_layout)
        if out_layout not in ("SMPL24", "AMASS"):
            raise ValueError(f"Unsupported output layout after adaptation: {out_layout}")
        frame_cfg.joints_category = out_layout
        # ONE host-to-device copy for coordinates and confidences (a pageable copy synchronises with the device's queue: two of
        # them were 0.19 ms of a 0.5 ms optimize_params_frame call)
        pts = np.asarray(pts, dtype=np.float32)
        cf = np.asarray(cf, dtype=np.float32)
        if pts.shape[:-1] == cf.shape and pts.shape[-1] == 3:
            both = torch.as_tensor(np.concatenate([pts, cf[..., None]], axis=-1), device=device)
            return both[..., :3].contiguous(), both[..., 3].contiguous(), None
        return (torch.as_tensor(pts, dtype=torch.float32, device=device),
                torch.as_tensor(cf, dtype=torch.float32, device=device), None)
    if joint_layout is not None and in_layout != "GENERIC":
        raise ValueError(
            "joint_layout adapters are currently defined for SMPL-family body "
            "layouts only. Use raw MANO/FLAME joint order with joint_layout=None.")
    frame_cfg.joints_category = "GENERIC"
    if model_indices is not None:
        model_indices = model_indices.to(dtype=torch.long)
    return xyz.to(device), conf.to(device), model_indices


def obtain_model(model, body_model: str, device):
    if model is None:
        model = load_body_model(BodyModelConfig(model_type=body_model), device)
    return as_body_model(model, device=device)


def check_param_type(params, body_model: str, name: str) -> None:
    expected = PARAM_TYPES[body_model]
    if not isinstance(params, expected):
        raise ValueError(f"{name} must be {expected.__name__} for body_model={body_model}.")
