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
                        confidence_input, device, pose_prior=None, dist=None):
    """Multi-frame shared-betas pre-pass (reference ``core/engine.py:217-262`` ->
    ``optimize_shape_multi_frame``, ``core/shape.py:10-115``).

    Loss over the first ``num_shape_frames`` frames at the mean pose: per frame
    ``sum_k conf_k^2 |(p_k - p_root)(beta) - (y_k - y_root)|^2 + w_s^2 |beta|^2`` (the per-frame
    translation is the root alignment, so it depends on beta through the rest root joint only).
    The reference runs ``torch.optim.LBFGS([betas], max_iter=num_shape_iters, lr=0.1,
    strong_wolfe)``; so does this function, with loss and gradient of every closure evaluation
    computed by ONE evaluate-only launch of the fused kernel over all frames (squared error via
    GMoF sigma -> inf, pose priors off) and the chain rule through the root alignment applied on
    the 10-vector.  The reference's Adam branch of this pass raises (``shape.py:10,110-113``:
    ``closure()`` under ``no_grad`` and no optimiser step), so ``use_lbfgs=False`` is rejected here
    with an explicit error instead.

    ``dist`` (an initialised ``torch.distributed`` module, one process per GPU) shards the ``num_shape_frames``
    frames over the ranks: every closure evaluation is then one launch over the rank's own block plus ONE
    all-reduce of NB + 4 floats (``parallel.allreduce_shape_terms``); all ranks run the same
    L-BFGS on the same reduced numbers and return the same betas.
    """
    if not seq_config.use_shape_optimization:
        return init_mean_shape
    if not seq_config.frame.use_lbfgs:
        raise RuntimeError(
            "use_shape_optimization=True with use_lbfgs=False: the reference's Adam branch of the shape "
            "pre-pass raises (core/shape.py:10,110-113); set use_shape_optimization=False or use_lbfgs=True")
    from .. import native
    from ..prior import MaxMixturePrior
    from .constants import category_indices, root_indices

    m = as_body_model(model)
    dev = m.device
    cat = seq_config.frame.joints_category
    smpl_index, corr_index = category_indices(cat)
    if smpl_index is None:
        raise ValueError(f"No such joints category: {cat}")
    root_model, root_target = root_indices(cat)
    t_size = data_tensor.shape[0]
    n = t_size if (seq_config.num_shape_frames < 0 or seq_config.num_shape_frames >= t_size) else seq_config.num_shape_frames
    y = torch.as_tensor(data_tensor, dtype=torch.float32).to(dev)[:n]
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        from ..parallel import shard_bounds
        lo, hi = shard_bounds(n, dist.get_world_size(), dist.get_rank())
        y = y[lo:hi]
        n = hi - lo                                            # may be 0 on trailing ranks of a short sequence
    else:
        dist = None
    targets = y[:, list(corr_index)].contiguous()
    conf = torch.as_tensor(confidence_input, dtype=torch.float32).to(dev)
    conf = (conf if conf.dim() == 1 else conf[0])[list(corr_index)].contiguous()
    pose = torch.as_tensor(init_mean_pose, dtype=torch.float32).to(dev).expand(n, -1).contiguous()
    go, bp = pose[:, :3].contiguous(), pose[:, 3:].contiguous()
    packed = getattr(m, "packed", False)             # SMPL-H / SMPL-X: all non-root joints in one pose vector, betas | expression
    if packed:
        bp = m.pack_pose(n, body_pose=bp[:, :3 * m.NUM_BODY_JOINTS]) if n > 0 else torch.zeros((0, 3 * (m.num_joints - 1)), device=dev)
    prior = pose_prior if pose_prior is not None else MaxMixturePrior(
        prior_folder="./data/models/", num_gaussians=seq_config.frame.pose_prior_num_gaussians, device=dev)
    jt, jd = m.native.joint_basis()                  # rest root joint = J_template[0] + J_dirs[0] . beta
    jt0 = torch.as_tensor(jt[root_model], device=dev)
    jd0 = torch.as_tensor(jd[root_model], device=dev)          # (3, NB)
    n_extra = int(jd0.shape[1]) - int(torch.as_tensor(init_mean_shape).reshape(-1).shape[0])   # expression coefficients: kept at 0
    if n_extra < 0:
        raise ValueError(f"init_mean_shape has more coefficients than the model's {int(jd0.shape[1])}")
    jd0 = jd0[:, :jd0.shape[1] - n_extra].contiguous()

    cfg = native.default_fit_config()
    cfg.num_iters, cfg.step_size = 1, 0.0
    cfg.sigma, cfg.joint_loss_weight = 1.0e8, 1.0             # plain squared error
    cfg.pose_prior_weight = cfg.angle_prior_weight = cfg.pose_preserve_weight = 0.0
    cfg.shape_prior_weight = float(seq_config.frame.shape_prior_weight)
    if packed:
        cfg.prior_pose_dims, cfg.num_betas_prior = 3 * m.NUM_BODY_JOINTS, m.num_betas

    betas = torch.as_tensor(init_mean_shape, dtype=torch.float32).to(dev).clone().reshape(1, -1).requires_grad_(True)

    from ..parallel import allreduce_shape_terms

    def closure():
        with torch.no_grad():
            b = betas.detach()
            nb = b.shape[1]
            if n > 0:
                transl = (y[:, root_target] - (jt0 + jd0 @ b[0])).contiguous()     # root alignment, (n,3)
                shape = b.expand(n, -1)
                if n_extra:
                    shape = torch.cat((shape, torch.zeros((n, n_extra), device=dev)), dim=1)
                r = native.fit_world(m.native, prior.native, cfg, list(smpl_index), targets, conf, go, bp,
                                     shape.contiguous(), transl, want_grad=True)
                g = r["grad"]
                g_beta = g[:, 3 + bp.shape[1]:3 + bp.shape[1] + nb].sum(dim=0)
                g_transl = g[:, 3 + bp.shape[1] + nb + n_extra:].sum(dim=0)
                loss = r["loss"].sum()
            else:
                g_beta, g_transl, loss = torch.zeros(nb, device=dev), torch.zeros(3, device=dev), torch.zeros((), device=dev)
            loss, g_beta, g_transl = allreduce_shape_terms(loss, g_beta, g_transl, dist)
            betas.grad = (g_beta - jd0.t() @ g_transl).reshape(1, -1)              # d transl / d beta = -J_dirs[root]
            return loss

    torch.optim.LBFGS([betas], max_iter=int(seq_config.num_shape_iters), lr=1e-1,
                      line_search_fn="strong_wolfe").step(closure)
    return betas.detach()
