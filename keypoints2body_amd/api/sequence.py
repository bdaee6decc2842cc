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
  (``num_iters_followup``, preserve term towards the shared initial pose) + ONE final forward.  When a
  ``torch.distributed`` process group is initialised (one process per GPU, ``torchrun``), the T frames are cut
  into contiguous blocks, one per rank (``parallel.fit_forward_exchange``: no collective during the iterations;
  the all-gather of the fitted parameters is enqueued under the rank's final forward over ITS block, the joints
  follow).  Every rank returns parameters, joints and loss of all frames; vertices stay on the rank that produced
  them (``vertices=None`` elsewhere) unless ``gather_vertices=True``.  Frame 0 is fitted once, by
  the rank that owns it.  ``bench.py`` times this very function.
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


def _packed_widths(est, prev: BodyModelParams):
    """Columns of the kernel-layout pose / shape vectors the fitter returns: for packed 52- / 55-joint models ALL non-root
    joints and betas | expression (162 / 153 and 20 values), else the widths of `prev` (63 / 69 and 10)."""
    smpl = getattr(est.fitter, "smpl", None)
    if smpl is not None and getattr(smpl, "packed", False):
        return 3 * (int(smpl.num_joints) - 1), int(smpl.num_shape)
    return int(prev.body_pose.shape[-1]), int(prev.betas.shape[-1])


def _fit_independent_frames(est, prev: BodyModelParams, xyz, conf, model_indices, dist, gather_vertices: bool = False):
    """The frame loop of ``api/sequence.py:214-281`` with ``use_previous_frame_init=False``: every frame starts from
    `prev`; frame 0 with ``seq_ind == 0`` semantics (``num_iters_first``, no preserve term), frames 1..T-1 with
    ``seq_ind >= 1`` semantics.  The T frames are cut into contiguous blocks, one per rank (one block = everything without a
    process group); a rank fits ITS block (frame 0 on the rank that owns it: one launch for it, one for the others), runs the
    final forward over ITS block only, and the ranks exchange the fitted parameters (enqueued under the forward) and the
    joints (``parallel.fit_forward_exchange``).  Vertices stay on the rank that produced them - ``vertices=None`` in the
    results of other ranks' frames - unless ``gather_vertices`` asks for the second exchange.

    Returns ``(params dict over all T frames, joints (T, .), vertex_of(i) -> (1, V, 3) | None, loss (T,))``."""
    from ..parallel import fit_forward_exchange, rows_per_rank, shard_bounds, unpack_outputs, valid_rows
    T = xyz.shape[0]
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist is not None else (1, 0)
    start, stop = shard_bounds(T, world, rank)
    per = rows_per_rank(T, world)
    pose_dim, num_shape = _packed_widths(est, prev)
    device = xyz.device

    def fit_block():
        parts = []
        if start == 0 and stop > 0:
            o, _, _, _ = est.fit_batch(_repeat_params(prev, 1), xyz[0:1], conf[0], seq_ind=0,
                                       target_model_indices=model_indices, per_frame_conf=False, run_forward=False)
            parts.append(o)
        lo = max(start, 1)
        if stop > lo:
            o, _, _, _ = est.fit_batch(_repeat_params(prev, stop - lo), xyz[lo:stop], conf[lo:stop], seq_ind=1,
                                       target_model_indices=model_indices, per_frame_conf=True, run_forward=False)
            parts.append(o)
        if not parts:                                     # more ranks than frames: this rank has no frame - zero rows of the right widths
            e = lambda c: torch.zeros((0, c), dtype=torch.float32, device=device)
            return {"global_orient": e(3), "body_pose": e(pose_dim), "betas": e(num_shape), "transl": e(3),
                    "loss": torch.zeros((0,), dtype=torch.float32, device=device)}
        if len(parts) == 1:
            return parts[0]
        return {k: torch.cat([p[k] for p in parts], dim=0).contiguous() for k in parts[0]}

    def forward_block(out):
        # (also for a rank without frames: the forward of zero rows returns zero-row joints / vertices, and the rank enters
        #  every collective of the exchange with padding only - never branch a collective on what a block holds)
        return est.fitter.final_forward(out)

    ex = fit_forward_exchange(fit_block, forward_block, dist, pad_to=per, gather_vertices=gather_vertices)
    if world == 1:
        out, joints, verts = ex["local"], ex["joints"], ex["vertices"]
        return out, joints, (lambda i: verts[i: i + 1]), out["loss"]
    keep = valid_rows(T, world, device)
    out = unpack_outputs(ex["packed"].index_select(0, keep), num_shape, pose_dim)
    joints, verts = ex["joints"], ex["vertices"]
    joints = joints.index_select(0, keep)
    if ex["vertices_gathered"]:
        verts = verts.index_select(0, keep)
        vertex_of = lambda i: verts[i: i + 1]
    else:
        vertex_of = lambda i: verts[i - start: i - start + 1] if start <= i < stop else None
    return out, joints, vertex_of, out["loss"]


def _repeat_params(p: BodyModelParams, n: int) -> BodyModelParams:
    """`p` (one row) as the start of n frames, every array field repeated and the data class kept - the reference hands
    `prev` itself (``SMPLHData`` / ``SMPLXData`` with hands, jaw, eyes, expression) to every frame (api/sequence.py:270-281)."""
    import dataclasses as dc
    changes = {}
    for f in dc.fields(p):
        v = getattr(p, f.name)
        if f.name == "metadata" or v is None or isinstance(v, (dict, str)):
            continue
        t = torch.as_tensor(v, dtype=torch.float32)
        if t.dim() == 1:
            t = t.unsqueeze(0)
        changes[f.name] = t.expand(n, -1).contiguous()
    return dc.replace(p, **changes)


def optimize_params_sequence(joints_seq, *, init_params: Optional[BodyModelParams] = None,
                             body_model: ModelType = "smpl", joint_layout: Optional[str] = None, model=None,
                             config: Optional[SequenceOptimizeConfig | dict] = None, device=None,
                             pose_prior=None, mean_params: Optional[tuple] = None,
                             gather_vertices: bool = False) -> list[BodyModelFitResult]:
    """Optimise body parameters for a motion sequence; results in temporal order.

    ``gather_vertices`` only matters for independent frames under a ``torch.distributed`` process group: every rank
    returns parameters, joints and loss of ALL frames, but the vertices of its own block only (``vertices=None``
    elsewhere) unless this asks for the second all-gather (82.7 KB per frame)."""
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
    if (seq_cfg.use_previous_frame_init and T > 1
            and hasattr(est.fitter, "chain_supported") and est.fitter.chain_supported(model_indices)):
        # world mode: the whole chain is one launch (Adam: k2b_fit_sequence, L-BFGS: k2b_fit_sequence_lbfgs); camera mode: the
        # stages are enqueued frame by frame, the reported loss and the final forward once for all frames
        if frame_cfg.coordinate_mode == "world" and prev.transl is None:
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
    if not hasattr(est.fitter, "fit_batch"):          # (a plug-in estimator without a batched entry point: one call per frame)
        for idx in range(T):
            results.append(engine.fit_frame(init_params=prev, j3d=xyz[idx: idx + 1], conf_3d=conf[idx], seq_ind=idx,
                                            target_model_indices=model_indices))
        return results
    out, joints, vertex_of, loss = _fit_independent_frames(est, prev, xyz, conf, model_indices, _process_group(),
                                                           gather_vertices)
    return [BodyModelFitResult(params=est.fitter.result_params(out, prev, slice(i, i + 1)), vertices=vertex_of(i),
                               joints=joints[i: i + 1], loss=loss[i]) for i in range(T)]


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
