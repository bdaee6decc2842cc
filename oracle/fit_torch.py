"""ORACLE (test infrastructure, not product code): CPU restatement of the
reference's world-space Adam fitting path in PyTorch (fp32, autograd).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.

Pinned by ``tests/golden/world_fit_*.npz``: fixtures produced by the reference's own
``WorldSpaceFitter.fit_frame`` (imported from ``/root/reference`` by
``oracle/gen_golden.py``) on seeded synthetic inputs; ``tests/test_oracle_fit.py``
checks this restatement against them.

Reference lines followed (paths relative to ``/root/reference/keypoints2body``):

* ``gmof``                          core/losses.py:6-10
* ``angle_prior``                   core/losses.py:13-21
* ``body_fitting_loss_3d``          core/losses.py:24-67
* ``MaxMixturePrior.__init__``      core/prior.py:98-176   (buffers)
* ``merged_log_likelihood``         core/prior.py:182-195
* ``WorldSpaceFitter.fit_frame``    core/fitters/world_space.py:93-323
  (clone/detach 121-125, conf handling 161-164, iteration count 214, parameter
  list 215-229, Adam loop 248-256, final forward 258-278)
* ``guess_init_transl_from_root``   core/fitters/world_space.py:13-50
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np
import torch

# Body-pose indices and signs of the elbow/knee bending prior (losses.py:16-19).
ANGLE_IDX = (52, 55, 9, 12)
ANGLE_SIGN = (1.0, -1.0, -1.0, -1.0)


@dataclass
class FitWeights:
    """Loss weights; defaults = what the reference's world fitter actually uses
    (losses.py:33-38 defaults, with joint/preserve weights injected from
    core/config.py:34-35 through estimators/optimization.py:75-85)."""

    sigma: float = 100.0
    pose_prior_weight: float = 4.78 * 1.5
    shape_prior_weight: float = 5.0
    angle_prior_weight: float = 15.2
    joint_loss_weight: float = 600.0
    pose_preserve_weight: float = 5.0


class GMMPrior:
    """Buffers of the max-mixture prior, built the way prior.py:133-163 builds them."""

    def __init__(self, means, covars, weights):
        covs32 = np.asarray(covars).astype(np.float32)
        self.means = torch.tensor(np.asarray(means).astype(np.float32))
        # prior.py:150-151: inverse taken per component on the float32 covariances
        self.precisions = torch.tensor(np.stack([np.linalg.inv(c) for c in covs32]).astype(np.float32))
        # prior.py:156-163: constant term from the *un-cast* covariances
        sqrdets = np.array([np.sqrt(np.linalg.det(c)) for c in np.asarray(covars)])
        const = (2 * np.pi) ** (69 / 2.0)
        nll = np.asarray(np.asarray(weights) / (const * (sqrdets / sqrdets.min())))
        self.nll_weights = torch.tensor(nll, dtype=torch.float32).unsqueeze(0)

    def per_component(self, pose: torch.Tensor) -> torch.Tensor:
        """(B,69) -> (B,M) values 0.5 d^T P d - log(nll_w)  (prior.py:183-189).  A shorter pose (SMPL-X: 63 body
        dimensions) is evaluated at [pose | 0 ...]: this engine's definition of the reference's inconsistent SMPL-X
        handling (SURVEY.md note N3); the goldens apply the same padding in front of the reference's own prior."""
        if pose.shape[1] < self.means.shape[1]:
            pose = torch.nn.functional.pad(pose, (0, self.means.shape[1] - pose.shape[1]))
        diff = pose.unsqueeze(1) - self.means
        pd = torch.einsum("mij,bmj->bmi", [self.precisions, diff])
        quad = (pd * diff).sum(dim=-1)
        return 0.5 * quad - torch.log(self.nll_weights)

    def __call__(self, pose: torch.Tensor, betas=None) -> torch.Tensor:
        vals, _ = torch.min(self.per_component(pose), dim=1)   # prior.py:194
        return vals


def gmof(x, sigma):
    x2 = x ** 2
    s2 = sigma ** 2
    return (s2 * x2) / (s2 + x2)


def frame_losses(body_pose, preserve_pose, betas, model_joints, j3d, prior: GMMPrior,
                 conf, w: FitWeights, preserve_on: bool) -> torch.Tensor:
    """Per-frame total loss (B,), term by term as losses.py:49-66."""
    if conf.dim() == 1:
        conf = conf.view(1, -1)
    joint = (w.joint_loss_weight ** 2) * ((conf ** 2) * gmof(model_joints - j3d, w.sigma).sum(dim=-1)).sum(dim=-1)
    pose_prior = (w.pose_prior_weight ** 2) * prior(body_pose, betas)
    signs = torch.tensor(ANGLE_SIGN, device=body_pose.device)
    angle = (w.angle_prior_weight ** 2) * (torch.exp(body_pose[:, list(ANGLE_IDX)] * signs) ** 2).sum(dim=-1)
    shape = (w.shape_prior_weight ** 2) * (betas ** 2).sum(dim=-1)
    wp = w.pose_preserve_weight if preserve_on else 0.0
    preserve = (wp ** 2) * ((body_pose - preserve_pose) ** 2).sum(dim=-1)
    return joint + pose_prior + angle + shape + preserve


@dataclass
class FitTrace:
    """Optional per-iteration record (parameters *after* the step, loss *before* it)."""

    iters: list = field(default_factory=list)
    global_orient: list = field(default_factory=list)
    body_pose: list = field(default_factory=list)
    betas: list = field(default_factory=list)
    transl: list = field(default_factory=list)
    loss: list = field(default_factory=list)          # (B,) per recorded iteration


@dataclass
class FitOutput:
    global_orient: torch.Tensor
    body_pose: torch.Tensor
    betas: torch.Tensor
    transl: torch.Tensor
    joints: torch.Tensor
    vertices: torch.Tensor
    loss: torch.Tensor            # (B,) loss of the last iteration, before its step
    trace: Optional[FitTrace] = None


def fit_world_adam(model, prior: GMMPrior, global_orient, body_pose, betas, transl, j3d,
                   conf=None, *, num_iters: int, lr: float = 1e-2, seq_ind: int = 0,
                   model_idx: Optional[Sequence[int]] = None, target_idx: Optional[Sequence[int]] = None,
                   weights: Optional[FitWeights] = None, freeze_betas: bool = False,
                   trace_iters: Sequence[int] = (), record_all_losses: bool = False) -> FitOutput:
    """Adam branch of ``WorldSpaceFitter.fit_frame`` for a batch of B frames.

    The loss is a sum over frames with no cross-frame term (losses.py:67) and Adam
    is element-wise, so a batch is B independent fits; the per-frame losses are kept
    apart here so a caller can sum them to the reference's scalar.
    """
    w = weights or FitWeights()
    go = global_orient.clone().detach().requires_grad_(True)
    bp = body_pose.clone().detach().requires_grad_(True)
    be = betas.clone().detach()
    be.requires_grad = not freeze_betas
    tr = transl.clone().detach().requires_grad_(True)
    preserve = bp.clone().detach()
    K = j3d.shape[1]
    if conf is None:
        conf = torch.ones(K)
    elif conf.dim() == 2:
        conf = conf[0]                     # world_space.py:163-164 (row 0 only)
    model_idx = list(range(K)) if model_idx is None else list(model_idx)
    target_idx = list(range(K)) if target_idx is None else list(target_idx)

    def per_frame():
        out = model(global_orient=go, body_pose=bp, betas=be, transl=tr)
        return frame_losses(bp, preserve, be, out.joints[:, model_idx, :], j3d[:, target_idx, :],
                            prior, conf[target_idx], w, preserve_on=seq_ind > 0)

    params = [go, bp, tr] + ([] if freeze_betas else [be])   # world_space.py:215-229
    opt = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999))
    trace = FitTrace() if (trace_iters or record_all_losses) else None
    last = None
    for it in range(1, num_iters + 1):
        opt.zero_grad()
        lf = per_frame()
        lf.sum().backward()
        opt.step()
        last = lf.detach()
        if trace is not None:
            if record_all_losses:
                trace.loss.append(last.clone())
            if it in trace_iters:
                trace.iters.append(it)
                trace.global_orient.append(go.detach().clone())
                trace.body_pose.append(bp.detach().clone())
                trace.betas.append(be.detach().clone())
                trace.transl.append(tr.detach().clone())
                if not record_all_losses:
                    trace.loss.append(last.clone())
    with torch.no_grad():
        out = model(global_orient=go, body_pose=bp, betas=be, transl=tr, return_full_pose=False)
    return FitOutput(go.detach(), bp.detach(), be.detach(), tr.detach(),
                     out.joints.detach(), out.vertices.detach(), last, trace)


SMPLX_FIELDS = ("global_orient", "body_pose", "transl", "left_hand_pose", "right_hand_pose", "expression", "jaw_pose",
                "leye_pose", "reye_pose", "betas")      # the optimiser's parameter list, world_space.py:215-229


SMPLH_FIELDS = ("global_orient", "body_pose", "transl", "left_hand_pose", "right_hand_pose", "betas")


def fit_world_adam_smplx(model, prior: GMMPrior, params: dict, j3d, conf=None, *, num_iters: int, lr: float = 1e-2,
                         seq_ind: int = 0, model_idx: Optional[Sequence[int]] = None, weights: Optional[FitWeights] = None,
                         freeze_betas: bool = False, trace_iters: Sequence[int] = (), fields: Sequence[str] = ()):
    """Adam branch of ``WorldSpaceFitter.fit_frame`` for ``SMPLXData`` inputs (world_space.py:126-151: hands, expression,
    jaw and eyes join the optimiser; 173-192: they are passed to the model; 202-212: the loss sees ``body_pose`` (63-D,
    prior zero-padded to the mixture's 69 dimensions - see ``GMMPrior.per_component``), ``betas`` (not the expression)
    and the model joints).  ``params``: dict of (B, .) tensors with the keys of ``SMPLX_FIELDS``.
    Returns (dict of fitted tensors, per-frame loss of the last iteration before its step, joints, vertices, trace)."""
    w = weights or FitWeights()
    # (``fields=SMPLH_FIELDS``: the same branch for ``SMPLHData`` - both hands join the optimiser, no face / expression)
    p = {k: params[k].clone().detach().requires_grad_(True) for k in (fields or SMPLX_FIELDS)}
    p["betas"].requires_grad = not freeze_betas
    preserve = p["body_pose"].clone().detach()
    K = j3d.shape[1]
    conf = torch.ones(K) if conf is None else (conf[0] if conf.dim() == 2 else conf)
    idx = list(range(K)) if model_idx is None else list(model_idx)

    def per_frame():
        out = model(**p)
        return frame_losses(p["body_pose"], preserve, p["betas"], out.joints[:, idx, :], j3d, prior, conf, w,
                            preserve_on=seq_ind > 0)

    opt = torch.optim.Adam([v for k, v in p.items() if v.requires_grad], lr=lr, betas=(0.9, 0.999))
    last, trace = None, {}
    for it in range(1, num_iters + 1):
        opt.zero_grad()
        lf = per_frame()
        lf.sum().backward()
        opt.step()
        last = lf.detach()
        if it in trace_iters:
            trace[it] = {k: v.detach().clone() for k, v in p.items()}
    with torch.no_grad():
        out = model(**p)
    return {k: v.detach() for k, v in p.items()}, last, out.joints.detach(), out.vertices.detach(), trace


def guess_init_transl(model, pose_aa, betas, j3d, root_model: int = 0, root_target: int = 0):
    """transl = target root - model root at the initial pose (world_space.py:34-50)."""
    with torch.no_grad():
        out = model(global_orient=pose_aa[:, :3], body_pose=pose_aa[:, 3:], betas=betas)
    return (j3d[:, root_target, :] - out.joints[:, root_model, :]).detach()


# ---------------------------------------------------------------------------------------------
# Camera-space two-stage fitter (reference core/fitters/camera_space.py:81-339,
# camera_fitting_loss_3d core/losses.py:70-93, guess_init_3d camera_space.py:16-41).
# Pinned by tests/golden/camera_fit_*.npz (oracle/gen_golden.py runs the reference class).
# The reference's broadcasts (losses.py:46-47, 91-93) are only valid for one frame per call, so
# this restatement fits ONE frame per call too; batches loop.
# ---------------------------------------------------------------------------------------------
TORSO = (2, 1, 17, 16)     # RHip, LHip, RShoulder, LShoulder in both SMPL24 and AMASS numbering


def guess_init_cam_t(model_joints, j3d):
    idx = list(TORSO)
    return (j3d[:, idx] - model_joints[:, idx]).sum(dim=1) / 4.0


def camera_stage1_loss(model_joints, cam_t, cam_t_est, j3d, depth_loss_weight=100.0):
    idx = list(TORSO)
    err = (j3d[:, idx] - (model_joints + cam_t)[:, idx]) ** 2            # (1,4,3)
    depth = (depth_loss_weight ** 2) * (cam_t - cam_t_est) ** 2           # (1,3): broadcast over the 4 joints
    return (err + depth).sum()


def fit_camera_adam_one(model, prior: GMMPrior, global_orient, body_pose, betas, j3d, conf=None, *, num_iters: int,
                        lr: float = 1e-2, seq_ind: int = 0, freeze_betas: bool = False,
                        weights: Optional[FitWeights] = None, model_idx: Optional[Sequence[int]] = None,
                        init_cam_t: Optional[torch.Tensor] = None) -> FitOutput:
    """Both Adam stages for ONE frame (all tensors have a leading dimension of 1)."""
    w = weights or FitWeights()
    go = global_orient.clone().detach()
    bp = body_pose.clone().detach()
    be = betas.clone().detach()
    K = j3d.shape[1]
    idx = list(range(K)) if model_idx is None else list(model_idx)
    with torch.no_grad():
        j0 = model(global_orient=go, body_pose=bp, betas=be).joints
    # camera_space.py:123-134: the caller may supply the initial translation (it is then also the
    # centre of the depth prior)
    t0 = guess_init_cam_t(j0, j3d).detach() if init_cam_t is None else init_cam_t.detach().clone()
    cam_t = t0.clone()
    preserve = bp.clone().detach()
    conf = torch.ones(K) if conf is None else conf

    go.requires_grad_(True)
    cam_t.requires_grad_(True)
    opt = torch.optim.Adam([go, cam_t], lr=lr, betas=(0.9, 0.999))
    for _ in range(num_iters):                                   # stage 1: camera_space.py:183-213
        joints = model(global_orient=go, body_pose=bp, betas=be).joints
        loss = camera_stage1_loss(joints[:, idx], cam_t, t0, j3d)
        opt.zero_grad()
        loss.backward()
        opt.step()
    stage1 = (go.detach().clone(), cam_t.detach().clone())

    bp.requires_grad_(True)
    opt_betas = seq_ind == 0 or not freeze_betas                 # camera_space.py:219-224
    be.requires_grad_(opt_betas)
    params = [bp] + ([be] if opt_betas else []) + [go, cam_t]
    opt = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999))
    for _ in range(num_iters):                                   # stage 2: camera_space.py:268-298
        joints = model(global_orient=go, body_pose=bp, betas=be).joints
        loss = frame_losses(bp, preserve, be, joints[:, idx] + cam_t, j3d, prior, conf, w,
                            preserve_on=seq_ind > 0).sum()
        opt.zero_grad()
        loss.backward()
        opt.step()
    with torch.no_grad():                                        # final: camera_space.py:300-326
        out = model(global_orient=go, body_pose=bp, betas=be, return_full_pose=False)
        wf = FitWeights(**{**w.__dict__, "joint_loss_weight": 600.0})
        final = frame_losses(bp, preserve, be, out.joints[:, idx] + cam_t, j3d, prior, conf, wf, preserve_on=False)
    res = FitOutput(go.detach(), bp.detach(), be.detach(), cam_t.detach(), out.joints.detach(),
                    out.vertices.detach(), final.detach())
    res.stage1 = stage1
    res.init_cam_t = t0
    return res


# ---------------------------------------------------------------------------------------------
# LBFGS branch of the world fitter (reference core/fitters/world_space.py:231-247) and the
# multi-frame shape pre-pass (reference core/shape.py:10-115).  Pinned by
# tests/golden/lbfgs_world_*.npz and tests/golden/shape_pass.npz.
# ---------------------------------------------------------------------------------------------
def fit_world_lbfgs_one(model, prior: GMMPrior, global_orient, body_pose, betas, transl, j3d, conf=None, *,
                        max_iter: int, lr: float = 1e-2, seq_ind: int = 0, freeze_betas: bool = False,
                        weights: Optional[FitWeights] = None, model_idx: Optional[Sequence[int]] = None) -> FitOutput:
    """One frame: ``torch.optim.LBFGS(params, max_iter, lr, strong_wolfe).step(closure)``, then the
    loss at the result."""
    w = weights or FitWeights()
    go = global_orient.clone().detach().requires_grad_(True)
    bp = body_pose.clone().detach().requires_grad_(True)
    tr = transl.clone().detach().requires_grad_(True)
    be = betas.clone().detach()
    be.requires_grad = not freeze_betas
    preserve = bp.clone().detach()
    K = j3d.shape[1]
    conf = torch.ones(K) if conf is None else conf
    idx = list(range(K)) if model_idx is None else list(model_idx)

    def total():
        out = model(global_orient=go, body_pose=bp, betas=be, transl=tr)
        return frame_losses(bp, preserve, be, out.joints[:, idx, :], j3d, prior, conf, w, preserve_on=seq_ind > 0).sum()

    params = [go, bp, tr] + ([] if freeze_betas else [be])
    opt = torch.optim.LBFGS(params, max_iter=max_iter, lr=lr, line_search_fn="strong_wolfe")

    def closure():
        opt.zero_grad()
        loss = total()
        loss.backward()
        return loss

    opt.step(closure)
    with torch.no_grad():
        final = total()
        out = model(global_orient=go, body_pose=bp, betas=be, transl=tr, return_full_pose=False)
    return FitOutput(go.detach(), bp.detach(), be.detach(), tr.detach(), out.joints.detach(), out.vertices.detach(),
                     final.detach().reshape(1))


def shape_pass_lbfgs(model, init_betas, pose_init, j3d_world, conf, frame_indices, *, num_iters: int = 40,
                     step_size: float = 1e-1, shape_prior_weight: float = 5.0, model_idx=None, root: int = 0):
    """Shared betas over several frames (shape.py:66-105, LBFGS branch): per frame the model is
    root-aligned to the target and the squared joint error plus ``w_s^2 |beta|^2`` is added."""
    betas = init_betas.clone().detach().requires_grad_(True)
    K = j3d_world.shape[1]
    idx = list(range(K)) if model_idx is None else list(model_idx)
    opt = torch.optim.LBFGS([betas], max_iter=num_iters, lr=step_size, line_search_fn="strong_wolfe")

    def closure():
        opt.zero_grad()
        total = betas.new_tensor(0.0)
        for t in frame_indices:
            pose_t = pose_init[t:t + 1]
            joints = model(global_orient=pose_t[:, :3], body_pose=pose_t[:, 3:], betas=betas).joints
            transl_t = j3d_world[t:t + 1, root, :] - joints[:, root, :]
            err = (joints + transl_t.unsqueeze(1))[:, idx, :] - j3d_world[t:t + 1]
            total = total + ((conf ** 2) * (err ** 2).sum(dim=-1)).sum() + (shape_prior_weight ** 2) * (betas ** 2).sum()
        total.backward()
        return total

    opt.step(closure)
    return betas.detach()
