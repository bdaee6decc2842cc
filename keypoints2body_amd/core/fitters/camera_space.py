"""Camera-space two-stage fitter on the HIP engine.

Drop-in for the reference's ``CameraSpaceFitter``
(reference ``keypoints2body/core/fitters/camera_space.py:44-339``).  Both stages run in the same
fused kernel as the world fitter (``k2b_fit_world``): the camera translation takes the place of
``transl`` (it is added to the joints inside the loss, ``core/losses.py:46-47``), and the stages
differ only in the fit configuration:

* stage 1 (``camera_space.py:183-213``): optimise ``[global_orient, camera_translation]``
  (``optimize_mask = 9``) against the four torso joints with a plain squared error
  (GMoF with sigma -> inf) plus the depth prior ``100^2 |t - t0|^2``, which the reference's
  broadcast adds once per torso joint (``losses.py:91-93``: effective weight ``200^2``);
  all pose / shape priors off (the mixture is skipped in-kernel);
* stage 2 (``camera_space.py:268-298``): ``body_fitting_loss_3d`` over all targets; betas are
  optimised iff ``seq_ind == 0 or not freeze_betas`` (``:219-224``); a fresh optimiser state;
* result (``:300-339``): model-space joints / vertices (no translation applied), ``params.transl``
  = camera translation, ``loss`` = the loss re-evaluated at the fitted parameters with joint
  weight 600 and no preserve term (an evaluate-only launch: one iteration with step size 0).

``use_lbfgs=True`` (``camera_space.py:144-182, 229-267``) keeps L-BFGS (strong Wolfe) as the outer algorithm of both
stages, as the reference does; every closure evaluation is one evaluate-only launch of the same kernel with the stage's
configuration (loss and analytic gradient, no autograd graph) and, since round 4, the optimiser's state machine runs on
the device as well (``k2b_fit_world_lbfgs``: one independent optimiser per frame; the stages' ``optimize_mask`` keeps the
parameters outside the optimiser fixed).  ``lbfgs_driver = "torch"`` selects the host-driven twin (``torch.optim.LBFGS``
itself, one frame at a time).

Vertex-selected joints among ``target_model_indices`` (model index >= 24) are handled inside ``k2b_fit_world`` in
both stages (two launches per iteration queued by the one call: fused kernel evaluate-only + vertex term with Adam tail).

Like the reference (whose broadcasts only hold for one frame per call) ``fit_frame`` treats every
frame independently; unlike it, any number of frames may be passed at once.
"""
from __future__ import annotations

from typing import Optional

import torch

from ... import native
from ...models.body_model import BodyModel, as_body_model
from ...models.smpl_data import BodyModelFitResult, SMPLData
from ...prior import MaxMixturePrior
from ..constants import JOINT_MAP, TORSO_JOINTS, category_indices

_TORSO_IDX = [JOINT_MAP[name] for name in TORSO_JOINTS]     # same indices in the AMASS numbering
_SQUARED_ERROR_SIGMA = 1.0e8                                # gmof(e, sigma) -> e^2 in fp32


def guess_init_3d(model_joints, j3d, joints_category="SMPL24", torso_index: Optional[torch.Tensor] = None,
                  torso_targets: Optional[torch.Tensor] = None):
    """Initial camera translation: mean offset of the four torso joints
    (reference ``camera_space.py:16-41``).  ``torso_index`` (the four indices as a tensor on the joints' device) and
    ``torso_targets`` (``j3d[:, torso]`` gathered before) spare the frame loop of a sequence an index upload per call; the
    arithmetic - and so the result, bit for bit - is the same."""
    if joints_category not in ("SMPL24", "AMASS"):
        raise ValueError(f"Unknown joints category: {joints_category}")
    if torso_index is None:
        return (j3d[:, _TORSO_IDX] - model_joints[:, _TORSO_IDX]).sum(dim=1) / 4.0
    tgt = torso_targets if torso_targets is not None else j3d.index_select(1, torso_index)
    return (tgt - model_joints.index_select(1, torso_index)).sum(dim=1) / 4.0


class CameraSpaceFitter:
    """Per-frame optimizer operating in camera coordinates, executed on one MI355X."""

    def __init__(self, smpl_model, step_size=1e-2, num_iters=100, use_lbfgs=True, joints_category="SMPL24",
                 device=None, pose_prior_num_gaussians=8, pose_prior: Optional[MaxMixturePrior] = None):
        self.smpl: BodyModel = as_body_model(smpl_model, device=device)
        self.device = self.smpl.device
        self.step_size = step_size
        self.num_iters = num_iters
        self.use_lbfgs = use_lbfgs
        self.joints_category = joints_category
        self.smpl_index, self.corr_index = category_indices(joints_category)
        self.pose_prior = pose_prior if pose_prior is not None else MaxMixturePrior(
            prior_folder="./data/models/", num_gaussians=pose_prior_num_gaussians, device=self.device)

    def _dev(self, x, cols) -> torch.Tensor:
        t = torch.as_tensor(x, dtype=torch.float32).detach().to(self.device)
        if t.dim() != 2 or t.shape[1] != cols:
            raise ValueError(f"expected a (B,{cols}) tensor, got {tuple(t.shape)}")
        return t.contiguous()

    def stage_configs(self, seq_ind, joint_loss_weight=600.0, pose_preserve_weight=5.0, freeze_betas=True,
                      depth_w=200.0):
        """Kernel configurations of the two stages: ``(cfg1, cfg2, fit_betas)``.

        Stage 1 (``camera_space.py:137-213``): ``[global_orient, camera_translation]`` (``optimize_mask = 9``),
        plain squared error on the stage's joints (GMoF sigma -> inf, weight 1), depth prior ``depth_w^2 |t - t0|^2``
        (200 = the reference's 100 broadcast over the four torso joints, ``losses.py:91-93``), every other prior off.
        Stage 2 (``:215-298``): ``body_fitting_loss_3d``; betas optimised iff ``seq_ind == 0 or not freeze_betas``."""
        cfg1 = native.default_fit_config()
        cfg1.num_iters, cfg1.step_size = int(self.num_iters), float(self.step_size)
        cfg1.sigma, cfg1.joint_loss_weight = _SQUARED_ERROR_SIGMA, 1.0
        cfg1.pose_prior_weight = cfg1.angle_prior_weight = cfg1.shape_prior_weight = cfg1.pose_preserve_weight = 0.0
        cfg1.optimize_mask, cfg1.transl_prior_weight = 9, float(depth_w)
        cfg2 = native.default_fit_config()
        cfg2.num_iters, cfg2.step_size = int(self.num_iters), float(self.step_size)
        cfg2.joint_loss_weight = float(joint_loss_weight)
        cfg2.pose_preserve_weight = float(pose_preserve_weight) if seq_ind > 0 else 0.0
        fit_betas = seq_ind == 0 or not freeze_betas
        cfg2.optimize_mask = 15 if fit_betas else 11
        return cfg1, cfg2, fit_betas

    def fit_batch(self, init_params: SMPLData, j3d: torch.Tensor, conf_3d: Optional[torch.Tensor] = None,
                  seq_ind: int = 0, target_model_indices: Optional[torch.Tensor] = None,
                  joint_loss_weight: float = 600.0, pose_preserve_weight: float = 5.0, freeze_betas: bool = True,
                  init_cam_t: Optional[torch.Tensor] = None, per_frame_conf: bool = False, want_vertices: bool = True,
                  run_forward: bool = True):
        """Both stages for B independent frames (two fused launches for the whole batch + one evaluate-only launch for the
        reported loss): ``(params dict of (B, .) tensors - ``transl`` = the camera translation -, joints, vertices,
        per-frame loss)``; ``run_forward=False`` leaves the final forward to the caller (``final_forward``), as the sharded
        sequence path wants it.  Every frame's result is that of its own single-frame ``fit_frame`` call, bit for bit."""
        go, bp, be = self._start_rows(init_params)
        j3d = torch.as_tensor(j3d, dtype=torch.float32).to(self.device)
        B = j3d.shape[0]
        if not (go.shape[0] == bp.shape[0] == be.shape[0] == B):
            raise ValueError("init_params and j3d disagree on the number of frames")
        tg = self._gather_targets(j3d, target_model_indices)
        model_idx, targets = tg["model_idx"], tg["targets"]
        conf = None if conf_3d is None else torch.as_tensor(conf_3d, dtype=torch.float32).to(self.device).contiguous()
        if conf is not None and conf.dim() == 2 and not per_frame_conf:
            conf = conf[0].contiguous()                                          # (the reference reads row 0 only)
        s2 = self._stages(go, bp, be, j3d, tg, conf, seq_ind, joint_loss_weight, pose_preserve_weight, freeze_betas, init_cam_t)

        out = {k: s2[k] for k in ("global_orient", "body_pose", "betas", "transl")}
        out["loss"] = self._loss_at(out, model_idx, targets, conf)
        if not run_forward:
            return out, None, None, out["loss"]
        joints, verts = self.final_forward(out, want_vertices=want_vertices)
        return out, joints, verts, out["loss"]

    def _start_rows(self, init_params):
        J = self.smpl.num_joints
        return (self._dev(init_params.global_orient, 3), self._dev(init_params.body_pose, 3 * (J - 1)),
                self._dev(init_params.betas, self.smpl.num_betas))

    def _device_index(self, idx) -> torch.Tensor:
        """`idx` as an int64 tensor on the device, uploaded once per index list."""
        cache = self.__dict__.setdefault("_index_cache", {})
        key = tuple(int(i) for i in idx)
        if key not in cache:
            cache[key] = torch.tensor(key, dtype=torch.int64, device=self.device)
        return cache[key]

    def _gather_targets(self, j3d, target_model_indices):
        """Targets of both stages for however many frames `j3d` holds: stage 2 fits ``model_idx`` to ``targets``, stage 1 the
        four torso joints (``camera_space.py:183-198``) or, with caller-chosen indices, all of them (``:199-210``)."""
        if target_model_indices is None:
            if self.smpl_index is None:
                raise ValueError("joints_category='GENERIC' needs target_model_indices")
            torso = self._device_index(_TORSO_IDX)
            identity = list(self.corr_index) == list(range(j3d.shape[1]))
            return dict(custom=False, model_idx=list(self.smpl_index),
                        targets=(j3d if identity else j3d.index_select(1, self._device_index(self.corr_index))).contiguous(),
                        stage1_idx=_TORSO_IDX, stage1_tgt=j3d.index_select(1, torso).contiguous(), depth_w=200.0, torso=torso)
        model_idx = [int(i) for i in torch.as_tensor(target_model_indices).reshape(-1).tolist()]
        targets = j3d.contiguous()
        return dict(custom=True, model_idx=model_idx, targets=targets, stage1_idx=model_idx, stage1_tgt=targets, depth_w=100.0,
                    torso=None)

    def _stages(self, go, bp, be, j3d, tg, conf, seq_ind, joint_loss_weight, pose_preserve_weight, freeze_betas,
                init_cam_t=None, rows: Optional[slice] = None):
        """Initial camera translation + both stages for the frames `rows` of `j3d` / `tg` (all of them by default), started
        from (go, bp, be) on the device: the parameters behind stage 2 (``transl`` = the camera translation).  Only launches
        are enqueued: nothing is uploaded or read back."""
        sl = slice(None) if rows is None else rows
        model_idx, targets = tg["model_idx"], tg["targets"][sl]
        stage1_idx, stage1_tgt = tg["stage1_idx"], tg["stage1_tgt"][sl]
        # initial camera translation (camera_space.py:110-134)
        if init_cam_t is None:
            if self.smpl.packed:
                joints0 = self.smpl(global_orient=go, body_pose=bp, betas=be, return_verts=False).joints
            else:
                joints0 = self.smpl.native.lbs(go, bp, be, None, want_vertices=False)[0]
            if not tg["custom"]:
                cam_t0 = guess_init_3d(joints0, None, self.joints_category, torso_index=tg["torso"], torso_targets=stage1_tgt)
            else:
                cam_t0 = j3d[sl][:, 0, :] - joints0[:, model_idx[0], :]
        else:
            cam_t0 = torch.as_tensor(init_cam_t, dtype=torch.float32).to(self.device)
        cam_t0 = cam_t0.detach().contiguous()

        def fit(cfg, idx, tgt, cf, p):
            # (vertex-selected joints among the targets are handled inside k2b_fit_world)
            cur = (p["global_orient"], p["body_pose"], p["betas"], p["transl"])
            cfg.conf_per_frame = int(cf is not None and cf.dim() == 2)
            return native.fit_world(self.smpl.native, self.pose_prior.native, cfg, idx, tgt, cf, *cur,
                                    transl_prior_target=cam_t0)

        cfg1, cfg2, fit_betas = self.stage_configs(seq_ind, joint_loss_weight, pose_preserve_weight, freeze_betas, tg["depth_w"])
        start = dict(global_orient=go, body_pose=bp, betas=be, transl=cam_t0)
        if self.use_lbfgs:
            return self._two_stages_lbfgs(cfg1, cfg2, fit_betas, stage1_idx, stage1_tgt, model_idx, targets, conf, start, cam_t0)
        s1 = fit(cfg1, stage1_idx, stage1_tgt, None, start)
        return fit(cfg2, model_idx, targets, conf, s1)

    def _loss_at(self, p, model_idx, targets, conf):
        """Loss at the fitted parameters: weight 600, no preserve term, no depth prior (``camera_space.py:316-326``) - one
        evaluate-only launch (one iteration with step size 0) over however many frames `p` holds."""
        cfg = native.default_fit_config()
        cfg.num_iters, cfg.step_size, cfg.joint_loss_weight = 1, 0.0, 600.0
        cfg.conf_per_frame = int(conf is not None and conf.dim() == 2)
        return native.fit_world(self.smpl.native, self.pose_prior.native, cfg, model_idx, targets, conf, p["global_orient"],
                                p["body_pose"], p["betas"], p["transl"])["loss"]

    def chain_supported(self, target_model_indices=None) -> bool:
        """Any target set: the stages go through ``k2b_fit_world`` / ``k2b_fit_world_lbfgs`` frame by frame."""
        return True

    def fit_chain(self, init_params: SMPLData, j3d: torch.Tensor, conf_3d: Optional[torch.Tensor] = None,
                  target_model_indices: Optional[torch.Tensor] = None, joint_loss_weight: float = 600.0,
                  pose_preserve_weight: float = 5.0, freeze_betas: bool = True, want_vertices: bool = True,
                  run_forward: bool = True):
        """One sequence in warm-start mode (the reference's default frame loop, ``api/sequence.py:214-281`` with
        ``use_previous_frame_init=True``, in camera mode): frame t starts from frame t-1's fitted pose and shape, its camera
        translation from ``guess_init_3d`` at that start (``camera_space.py:110-134``; the previous translation is not used).

        The two stages of a frame depend on the frame before, so they are enqueued frame by frame (nothing is read back in
        between); what does NOT depend on the next frame leaves the loop: the loss at the fitted parameters is ONE
        evaluate-only launch over all T frames and the final forward ONE LBS call, instead of one of each per frame.
        Row t of every result is what the reference's loop of ``fit_frame`` calls returns for frame t, bit for bit with this
        fitter's own ``fit_frame`` loop (``tests/test_gpu_api.py``)."""
        j3d = torch.as_tensor(j3d, dtype=torch.float32).to(self.device)
        T = j3d.shape[0]
        conf = None if conf_3d is None else torch.as_tensor(conf_3d, dtype=torch.float32).to(self.device).contiguous()
        go, bp, be = self._start_rows(init_params)
        if not (go.shape[0] == bp.shape[0] == be.shape[0] == 1):
            raise ValueError("fit_chain starts from ONE row of parameters")
        tg = self._gather_targets(j3d, target_model_indices)             # both stages' targets of all T frames, gathered once
        keys, rows = ("global_orient", "body_pose", "betas", "transl"), []
        for t in range(T):
            cf = None if conf is None else (conf[t] if conf.dim() == 2 else conf)
            o = self._stages(go, bp, be, j3d, tg, cf, t, joint_loss_weight, pose_preserve_weight, freeze_betas,
                             rows=slice(t, t + 1))
            rows.append(o)
            go, bp, be = o["global_orient"], o["body_pose"], o["betas"]
        out = {k: torch.cat([r[k] for r in rows], dim=0).contiguous() for k in keys}
        out["loss"] = self._loss_at(out, tg["model_idx"], tg["targets"], conf)
        if not run_forward:
            return out, None, None, out["loss"]
        joints, verts = self.final_forward(out, want_vertices=want_vertices)
        return out, joints, verts, out["loss"]

    def final_forward(self, out, want_vertices=True):
        """Final forward of the reference (camera_space.py:300-314): model-space joints / vertices - the camera translation
        is part of the parameters, not applied to the mesh."""
        return self.smpl.native.lbs(out["global_orient"], out["body_pose"], out["betas"], None, want_vertices=want_vertices)

    def result_params(self, out, init_params=None, rows: Optional[slice] = None):
        sl = slice(None) if rows is None else rows
        return SMPLData(betas=out["betas"][sl], global_orient=out["global_orient"][sl], body_pose=out["body_pose"][sl],
                        transl=out["transl"][sl])

    def fit_frame(self, init_params: SMPLData, j3d: torch.Tensor, conf_3d: Optional[torch.Tensor] = None,
                  seq_ind: int = 0, target_model_indices: Optional[torch.Tensor] = None,
                  joint_loss_weight: float = 600.0, pose_preserve_weight: float = 5.0, freeze_betas: bool = True,
                  init_cam_t: Optional[torch.Tensor] = None) -> BodyModelFitResult:
        out, joints, verts, loss = self.fit_batch(init_params, j3d, conf_3d, seq_ind, target_model_indices, joint_loss_weight,
                                                  pose_preserve_weight, freeze_betas, init_cam_t=init_cam_t)
        return BodyModelFitResult(params=self.result_params(out), vertices=verts, joints=joints, loss=loss.sum())

    # ------------------------------------------------------------------------------------------------
    def _two_stages_lbfgs(self, cfg1, cfg2, fit_betas, idx1, tgt1, idx2, tgt2, conf, start, cam_t0):
        """LBFGS branches of both stages, frame by frame (``camera_space.py:144-182`` and ``:229-267``):
        ``torch.optim.LBFGS(params, max_iter=num_iters, lr=step_size, line_search_fn="strong_wolfe")`` with
        loss and gradient of every closure call from an evaluate-only launch."""
        max_iter, lr = int(self.num_iters), float(self.step_size)
        B, D = start["global_orient"].shape[0], start["body_pose"].shape[1]
        NB = start["betas"].shape[1]
        preserve = start["body_pose"].clone()                     # camera_space.py:136
        if getattr(self, "lbfgs_driver", "device") == "device":
            # both stages on the device (k2b_fit_world_lbfgs): the stages' optimize_mask keeps the parameters outside the
            # optimiser fixed (stage 1: global_orient + camera translation, camera_space.py:142; stage 2: :219-224)
            cur = start
            for cfg, idx, tgt, cf in ((cfg1, idx1, tgt1, None), (cfg2, idx2, tgt2, conf)):
                cfg.conf_per_frame = int(cf is not None and cf.dim() == 2)
                cur = native.fit_world_lbfgs(self.smpl.native, self.pose_prior.native, cfg, idx, tgt, cf, cur["global_orient"],
                                             cur["body_pose"], cur["betas"], cur["transl"], max_iter=max_iter, lr=lr,
                                             preserve_pose=preserve, transl_prior_target=cam_t0)
            return {k: cur[k] for k in ("global_orient", "body_pose", "betas", "transl")}
        for cfg in (cfg1, cfg2):
            cfg.num_iters, cfg.step_size = 1, 0.0
        rows = {k: [] for k in ("global_orient", "body_pose", "betas", "transl")}
        # (the optimiser's own arithmetic runs on HOST tensors, as in the reference: see WorldSpaceFitter._fit_lbfgs)
        dev = self.device
        order = ("global_orient", "body_pose", "betas", "transl")
        for f in range(B):
            sl = slice(f, f + 1)
            p = {k: start[k][sl].detach().to("cpu").clone() for k in rows}
            cf = conf[sl].contiguous() if (conf is not None and conf.dim() == 2) else conf
            pres_f, cam_f = preserve[sl].contiguous(), cam_t0[sl].contiguous()

            def run(cfg, idx, tgt, cfv, opt_keys):
                params = [p[k].requires_grad_(True) for k in opt_keys]
                cols = {"global_orient": slice(0, 3), "body_pose": slice(3, 3 + D), "betas": slice(3 + D, 3 + D + NB),
                        "transl": slice(3 + D + NB, 3 + D + NB + 3)}
                tgt_f = tgt[sl].contiguous()

                def closure():
                    with torch.no_grad():
                        flat = torch.cat([p[k].detach() for k in order], dim=1).to(dev)
                        cur = tuple(flat[:, cols[k]].contiguous() for k in order)
                        r = native.fit_world(self.smpl.native, self.pose_prior.native, cfg, idx, tgt_f, cfv, *cur,
                                             preserve_pose=pres_f, want_grad=True, transl_prior_target=cam_f)
                        back = torch.cat((r["grad"], r["loss"][:, None]), dim=1).cpu()
                    for k in opt_keys:
                        p[k].grad = back[:, cols[k]].clone()
                    return back[:, -1].sum()

                torch.optim.LBFGS(params, max_iter=max_iter, lr=lr, line_search_fn="strong_wolfe").step(closure)
                for k in opt_keys:
                    p[k] = p[k].detach()

            run(cfg1, idx1, tgt1, None, ["global_orient", "transl"])                       # camera_space.py:142
            run(cfg2, idx2, tgt2, cf, ["body_pose"] + (["betas"] if fit_betas else []) + ["global_orient", "transl"])  # :219-224
            for k in rows:
                rows[k].append(p[k].detach().to(dev))
        return {k: torch.cat(v, dim=0).contiguous() for k, v in rows.items()}
