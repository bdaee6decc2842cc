"""``optimize_params_sequence`` / ``optimize_shape_sequence`` (reference
``keypoints2body/api/sequence.py:40-319``), executed on the HIP engine.

Two execution modes, both with the reference's per-frame semantics:

* ``use_previous_frame_init=True`` (reference default): every frame starts from the previous
  frame's result, an inherently sequential chain (``api/sequence.py:280-281``).  World mode, Adam branch,
  kinematic targets (SMPL and SMPL-X): the WHOLE chain is one launch (``k2b_fit_sequence``: the frame loop runs
  inside the kernel, parameters and optimiser state never leave registers) followed by one final forward over
  all frames; any other configuration: one single-frame call per frame;
* ``use_previous_frame_init=False``: every frame starts from the same initial parameters, so
  the frames are independent and are fitted in TWO launches: frame 0
  (``num_iters_first``, no preserve term) and frames 1..T-1 as one batch
  (``num_iters_followup``, preserve term towards the shared initial pose).  When a
  ``torch.distributed`` process group is initialised (one process per GPU, ``torchrun``), that batch
  is SHARDED over the ranks (``parallel.fit_frames_sharded``: contiguous blocks, no collective during
  the iterations, one all-gather of the fitted parameters); every rank then runs the cheap final
  forward over the whole sequence and returns the full, identical list of results.
"""
from __future__ import annotations

from typing import Optional

import torch

from ..core.config import ModelType, SequenceOptimizeConfig, sequence_config_from
from ..core.engine import (OptimizeEngine, default_init_params, load_mean_pose_shape, optimize_shape_pass,
                           upgrade_smpl_family_init_params)
from ..core.joints.adapters import normalize_sequence_observations
from ..models.smpl_data import BodyModelFitResult, BodyModelParams, SMPLData
from . import common
from .frame import _with_root_aligned_transl


def _process_group():
    """The ``torch.distributed`` module when this process is one rank of an initialised group of > 1, else None."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


def _fit_independent_frames(est, prev: BodyModelParams, xyz, conf, model_indices, dist):
    """Frames that all start from `prev` (seq_ind >= 1 semantics), fitted in one batched launch or, under a
    process group, in one launch per rank over its contiguous block + ONE all-gather of the parameters
    (reference seam: the frame loop of ``api/sequence.py:214-281`` with ``use_previous_frame_init=False``)."""
    n = xyz.shape[0]
    if dist is None:
        return est.fit_batch(_repeat_params(prev, n), xyz, conf, seq_ind=1, target_model_indices=model_indices,
                             per_frame_conf=True)
    from ..parallel import fit_frames_sharded
    pose_dim, num_betas = int(prev.body_pose.shape[-1]), int(prev.betas.shape[-1])
    device = xyz.device

    def fit_block(sl: slice):
        b = sl.stop - sl.start
        if b <= 0:
            e = lambda c: torch.zeros((0, c), dtype=torch.float32, device=device)
            return {"global_orient": e(3), "body_pose": e(pose_dim), "betas": e(num_betas), "transl": e(3),
                    "loss": torch.zeros((0,), dtype=torch.float32, device=device)}
        out, _, _, _ = est.fit_batch(_repeat_params(prev, b), xyz[sl], conf[sl], seq_ind=1,
                                     target_model_indices=model_indices, per_frame_conf=True, run_forward=False)
        return out

    out = fit_frames_sharded(fit_block, n, num_betas, pose_dim, dist)
    joints, verts = est.fitter.final_forward(out)        # every rank: whole sequence, identical bits
    return out, joints, verts, out["loss"]


def _repeat_params(p: BodyModelParams, n: int) -> SMPLData:
    rep = lambda x: torch.as_tensor(x, dtype=torch.float32).expand(n, -1).contiguous()
    return SMPLData(betas=rep(p.betas), global_orient=rep(p.global_orient), body_pose=rep(p.body_pose),
                    transl=rep(p.transl))


def optimize_params_sequence(joints_seq, *, init_params: Optional[BodyModelParams] = None,
                             body_model: ModelType = "smpl", joint_layout: Optional[str] = None, model=None,
                             config: Optional[SequenceOptimizeConfig | dict] = None, device=None,
                             pose_prior=None, mean_params: Optional[tuple] = None) -> list[BodyModelFitResult]:
    """Optimise body parameters for a motion sequence; results in temporal order."""
    device = common.resolve_device(device)
    seq_cfg = sequence_config_from(config)
    frame_cfg = seq_cfg.frame
    common.check_request(frame_cfg, body_model)
    xyz, conf, model_indices, in_layout = normalize_sequence_observations(joints_seq, layout=joint_layout,
                                                                         body_model=body_model)
    xyz, conf, model_indices = common.canonicalize(xyz, conf, model_indices, in_layout, joint_layout, body_model,
                                                   frame_cfg, device)
    if seq_cfg.limit_frames is not None and seq_cfg.limit_frames > 0:
        xyz, conf = xyz[: seq_cfg.limit_frames], conf[: seq_cfg.limit_frames]
    if seq_cfg.fix_foot and xyz.shape[1] > 11:       # api/sequence.py:124-128
        conf = conf.clone()
        conf[:, [7, 8, 10, 11]] = 1.5

    model = common.obtain_model(model, body_model, device)
    mean_pose, mean_shape = mean_params if mean_params is not None else load_mean_pose_shape(
        common.DEFAULT_MEAN_FILE, device)
    mean_pose, mean_shape = mean_pose.to(device), mean_shape.to(device)
    betas_opt = mean_shape
    if frame_cfg.joints_category != "GENERIC":
        betas_opt = optimize_shape_pass(model=model, seq_config=seq_cfg, init_mean_shape=mean_shape,
                                        init_mean_pose=mean_pose, data_tensor=xyz, confidence_input=conf[0],
                                        device=device, pose_prior=pose_prior)
    engine = OptimizeEngine(model=model, frame_config=frame_cfg, device=device, model_type=body_model,
                            pose_prior=pose_prior)
    if xyz.shape[0] == 0:
        return []

    if init_params is None:
        base = default_init_params(mean_pose, betas_opt, xyz[0:1], model, joints_category=frame_cfg.joints_category,
                                   coordinate_mode=frame_cfg.coordinate_mode)
        prev = upgrade_smpl_family_init_params(base, model_type=body_model, model=model, device=device)
    else:
        common.check_param_type(init_params, body_model, "init_params")
        prev = init_params.to(device)

    results: list[BodyModelFitResult] = []
    T = xyz.shape[0]
    est = engine.estimator
    if (seq_cfg.use_previous_frame_init and T > 1 and frame_cfg.coordinate_mode == "world"
            and hasattr(est.fitter, "chain_supported") and est.fitter.chain_supported(model_indices)):
        if prev.transl is None:
            prev = _with_root_aligned_transl(prev, xyz[0:1], model, frame_cfg, device)
        out, joints, verts, loss = est.fit_chain(prev, xyz, conf, model_indices)
        return _batched_results(est, out, joints, verts, loss, prev, T)
    if seq_cfg.use_previous_frame_init or T == 1:
        for idx in range(T):
            frame = xyz[idx: idx + 1]
            if frame_cfg.coordinate_mode == "world" and prev.transl is None:
                prev = _with_root_aligned_transl(prev, frame, model, frame_cfg, device)
            res = engine.fit_frame(init_params=prev, j3d=frame, conf_3d=conf[idx], seq_ind=idx,
                                   target_model_indices=model_indices)
            results.append(res)
            if seq_cfg.use_previous_frame_init:
                prev = res.params
        return results

    # independent frames: all start from `prev` (api/sequence.py:214-281 with use_previous_frame_init=False)
    if frame_cfg.coordinate_mode == "world" and prev.transl is None:
        prev = _with_root_aligned_transl(prev, xyz[0:1], model, frame_cfg, device)
    results.append(engine.fit_frame(init_params=prev, j3d=xyz[0:1], conf_3d=conf[0], seq_ind=0,
                                    target_model_indices=model_indices))
    if not hasattr(est.fitter, "fit_batch"):          # camera-space fitter: one call per frame
        for idx in range(1, T):
            results.append(engine.fit_frame(init_params=prev, j3d=xyz[idx: idx + 1], conf_3d=conf[idx], seq_ind=idx,
                                            target_model_indices=model_indices))
        return results
    out, joints, verts, loss = _fit_independent_frames(est, prev, xyz[1:], conf[1:], model_indices, _process_group())
    return results + _batched_results(est, out, joints, verts, loss, prev, T - 1)


def _batched_results(est, out, joints, verts, loss, init, n) -> list[BodyModelFitResult]:
    """Per-frame result objects (the data class the reference returns for the model / input type) of a batched fit."""
    return [BodyModelFitResult(params=est.fitter.result_params(out, init, slice(i, i + 1)), vertices=verts[i: i + 1],
                               joints=joints[i: i + 1], loss=loss[i]) for i in range(n)]


def optimize_shape_sequence(joints_seq, *, body_model: ModelType = "smpl", joint_layout: Optional[str] = None,
                            model=None, config: Optional[SequenceOptimizeConfig | dict] = None, device=None,
                            pose_prior=None, mean_params: Optional[tuple] = None) -> BodyModelParams:
    """Run the sequence optimisation and return the last frame's parameters."""
    results = optimize_params_sequence(joints_seq, init_params=None, body_model=body_model, joint_layout=joint_layout,
                                       model=model, config=config, device=device, pose_prior=pose_prior,
                                       mean_params=mean_params)
    if not results:
        raise ValueError("No frames were optimized")
    return results[-1].params
