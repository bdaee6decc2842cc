"""World-space per-frame fitter on the HIP engine.

Drop-in for the reference's ``WorldSpaceFitter`` (reference
``keypoints2body/core/fitters/world_space.py:53-323``): same constructor and
``fit_frame`` signature, same result object.  Where the reference builds a loss
closure and steps ``torch.optim.Adam`` in Python (``world_space.py:248-256``), this
class validates its inputs, uploads them and makes ONE call into ``libk2b.so``
(``k2b_fit_world``: all iterations fused in one kernel, one frame per wavefront) plus
one ``k2b_lbs`` call for the final vertices/joints (``world_space.py:258-278``).

``use_lbfgs=True`` (the reference's default, ``world_space.py:231-247``) keeps L-BFGS with the strong-Wolfe line search
as the outer algorithm, exactly as the reference does, but its closure no longer builds an autograd graph: loss and
gradient of every evaluation come from an evaluate-only launch of the same kernel (``step_size = 0``, ``grad_out``).
Since round 4 the optimiser itself runs on the device too (``k2b_fit_world_lbfgs``, ``csrc/k2b_lbfgs.hip``: torch's
``LBFGS.step`` / ``_strong_wolfe`` restated as a per-frame state machine; the host only queues launches).  L-BFGS couples
all parameters it is given: the reference hands ONE optimiser the parameters of the whole batch, so for B > 1 its frames
share a line search; the API only ever calls it with B = 1 (``api/sequence.py:215``).  This class runs one independent
optimiser per frame, which equals the reference for B = 1 and deliberately differs (independent frames) for B > 1.
``lbfgs_driver = "host" | "torch"`` select the host-driven twins (``core/lbfgs_batched.py``; ``torch.optim.LBFGS`` itself).

Differences, all deliberate and documented in DESIGN.md:

* SMPL-H (52-joint models, ``SMPLHData``: both hands) and SMPL-X (55-joint models, ``SMPLXData``): hands, jaw, eyes and expression join the optimiser exactly as in the
  reference (``world_space.py:126-151,215-229``), fitted by the tree kernel (``csrc/k2b_fit_tree.hip``) over one packed
  pose / shape vector.  The reference hands SMPL-X's 63-D body pose to its 69-D mixture, which raises;
  here the mixture is evaluated at ``[body_pose | 0 x 6]``, and hands are full axis-angle poses (``use_pca=False``).
  Both branches: Adam fused, L-BFGS over evaluate-only launches of the tree kernel (one optimiser over the packed vectors:
  L-BFGS only ever forms inner products of the flattened parameters, so the packing does not change it);
  With a 24-joint model, hand / face fields of ``SMPLHData`` / ``SMPLXData`` inputs are carried through unchanged;
* vertex-selected joints (model joint index >= 24: smplx's "extra" joints, single mesh vertices): the fused
  kernel fits kinematic joints only, so ``k2b_fit_world`` then queues two launches per iteration (its kernel in
  evaluate-only mode for the kinematic targets and the priors, the vertex-term kernel with an Adam tail for the
  vertex targets and the step) - still one C call per fit, no host work between the launches, per-frame
  confidences allowed (the LBFGS branch gets loss and gradient of both terms from one evaluate-only call);
* ``fit_batch`` fits B independent frames in one launch with optional per-frame
  confidences; ``fit_frame`` keeps the reference's behaviour of using row 0 of a 2-D
  confidence tensor (``world_space.py:163-164``).
"""
from __future__ import annotations

import dataclasses as _dc
from typing import Optional

import numpy as np
import torch

from ... import native
from ...models.body_model import BodyModel, as_body_model
from ...models.smpl_data import BodyModelFitResult, SMPLData, SMPLHData, SMPLXData
from ...prior import MaxMixturePrior
from ..constants import category_indices, root_indices


def guess_init_transl_from_root(smpl_model, pose_aa, betas, j3d_world_frame, joints_category="SMPL24"):
    """Initial translation = target root - model root at the initial pose
    (reference ``core/fitters/world_space.py:13-50``).  One joints-only LBS launch."""
    model = as_body_model(smpl_model)
    root_model, root_target = root_indices(joints_category)
    pose_aa = torch.as_tensor(pose_aa, dtype=torch.float32)
    out = model(global_orient=pose_aa[:, :3], body_pose=pose_aa[:, 3:], betas=betas, return_verts=False)
    target = torch.as_tensor(j3d_world_frame, dtype=torch.float32).to(model.device)
    return (target[:, root_target, :] - out.joints[:, root_model, :]).detach()


class WorldSpaceFitter:
    """Per-frame optimizer operating in world coordinates, executed on one MI355X."""

    def __init__(self, smpl_model, step_size=1e-2, num_iters_first=30, num_iters_followup=10, use_lbfgs=True,
                 joints_category="SMPL24", device=None, pose_prior_num_gaussians=8,
                 pose_prior: Optional[MaxMixturePrior] = None):
        self.smpl: BodyModel = as_body_model(smpl_model, device=device)
        if self.smpl.num_joints not in (24, 52, 55):
            raise NotImplementedError(
                f"a body model with {self.smpl.num_joints} joints: the fit kernels are built for the 24-joint SMPL tree "
                "(SMPL-H / SMPL-X parameter sets ride on it unfitted), the 52-joint SMPL-H and the 55-joint SMPL-X tree")
        self.device = self.smpl.device
        self.step_size = step_size
        self.num_iters_first = num_iters_first
        self.num_iters_followup = num_iters_followup
        self.use_lbfgs = use_lbfgs
        self.joints_category = joints_category
        self.smpl_index, self.corr_index = category_indices(joints_category)   # raises on unknown category
        # the reference loads ./data/models/gmm_XX.pkl relative to the CWD (world_space.py:87-91)
        self.pose_prior = pose_prior if pose_prior is not None else MaxMixturePrior(
            prior_folder="./data/models/", num_gaussians=pose_prior_num_gaussians, device=self.device)

    # ------------------------------------------------------------------------------------
    def _config(self, seq_ind, joint_loss_weight, pose_preserve_weight, freeze_betas, per_frame_conf):
        cfg = native.default_fit_config()
        cfg.num_iters = int(self.num_iters_first if seq_ind == 0 else self.num_iters_followup)
        cfg.step_size = float(self.step_size)
        cfg.joint_loss_weight = float(joint_loss_weight)
        cfg.pose_preserve_weight = float(pose_preserve_weight) if seq_ind > 0 else 0.0   # world_space.py:211
        cfg.freeze_betas = int(bool(freeze_betas))
        cfg.conf_per_frame = int(bool(per_frame_conf))
        return cfg

    def _dev(self, x, cols=None) -> torch.Tensor:
        t = torch.as_tensor(x, dtype=torch.float32).detach().to(self.device)
        if cols is not None and (t.dim() != 2 or t.shape[1] != cols):
            raise ValueError(f"expected a (B,{cols}) tensor, got {tuple(t.shape)}")
        return t.contiguous()

    def _prepare(self, init_params, j3d, conf_3d, target_model_indices, per_frame_conf, num_init=None):
        """Argument handling shared by ``fit_batch`` and ``fit_chain``: device tensors in kernel layout, the joint
        gather of world_space.py:194-201 and the confidence rules of world_space.py:163-164.  ``num_init``: expected
        rows of ``init_params`` (default: one per frame of ``j3d``)."""
        if init_params.transl is None:
            raise ValueError("init_params.transl must be provided")
        j3d = torch.as_tensor(j3d, dtype=torch.float32)
        if j3d.dim() != 3 or j3d.shape[2] != 3:
            raise ValueError(f"j3d must be (B,K,3), got {tuple(j3d.shape)}")
        J = self.smpl.num_joints
        B = j3d.shape[0]
        smplx = self.smpl.packed                         # 52- / 55-joint trees (SMPL-H / SMPL-X)
        go = self._dev(init_params.global_orient, 3)
        if smplx:                                        # one pose vector of all non-root joints, one of all shape coefficients
            get = lambda name: getattr(init_params, name, None)
            bp = self.smpl.pack_pose(go.shape[0], body_pose=init_params.body_pose, jaw_pose=get("jaw_pose"),
                                     leye_pose=get("leye_pose"), reye_pose=get("reye_pose"),
                                     left_hand_pose=get("left_hand_pose"), right_hand_pose=get("right_hand_pose"))
            be = self.smpl.pack_shape(go.shape[0], init_params.betas, get("expression"))
        else:
            bp = self._dev(init_params.body_pose, 3 * (J - 1))
            be = self._dev(init_params.betas, self.smpl.num_betas)
        tr = self._dev(init_params.transl, 3)
        if not (go.shape[0] == bp.shape[0] == be.shape[0] == tr.shape[0] == (B if num_init is None else num_init)):
            raise ValueError("init_params and j3d disagree on the number of frames")

        # joint gather (world_space.py:194-201): model joint per target, and the target rows
        if target_model_indices is None:
            if self.smpl_index is None:
                raise ValueError("joints_category='GENERIC' needs target_model_indices")
            model_idx = list(self.smpl_index)
            conf_sel = list(self.corr_index)
            if conf_sel == list(range(j3d.shape[1])):    # AMASS / SMPL24 inputs arrive in target order: the gather is the
                tgt, conf_sel = j3d, None                # identity (on a device tensor it would upload its index list per call)
            else:
                tgt = j3d[:, conf_sel, :]
        else:
            model_idx = [int(i) for i in torch.as_tensor(target_model_indices).reshape(-1).tolist()]
            if len(model_idx) != j3d.shape[1]:
                raise ValueError("target_model_indices must have one entry per target joint")
            tgt = j3d
            conf_sel = None
        tgt = tgt.to(self.device).contiguous()
        if conf_3d is not None:
            conf = torch.as_tensor(conf_3d, dtype=torch.float32)
            if conf.dim() == 2 and not per_frame_conf:
                conf = conf[0]                           # reference quirk, world_space.py:163-164
            if conf_sel is not None:
                conf = conf[..., conf_sel]
            conf = conf.to(self.device).contiguous()
        else:
            conf = None

        return go, bp, be, tr, model_idx, tgt, conf

    def fit_batch(self, init_params: SMPLData, j3d, conf_3d=None, seq_ind: int = 0, target_model_indices=None,
                  joint_loss_weight: float = 600.0, pose_preserve_weight: float = 5.0, freeze_betas: bool = False,
                  per_frame_conf: bool = False, want_vertices: bool = True, run_forward: bool = True):
        """Fit B independent frames in one launch.

        Returns ``(params: dict of (B,.) tensors, joints, vertices, per_frame_loss)``; with
        ``run_forward=False`` the final forward is left to the caller (``final_forward``) and joints / vertices
        are ``None`` - the sharded sequence path gathers the parameters first.
        """
        go, bp, be, tr, model_idx, tgt, conf = self._prepare(init_params, j3d, conf_3d, target_model_indices, per_frame_conf)
        J = self.smpl.num_joints
        smplx = self.smpl.packed
        cfg = self._config(seq_ind, joint_loss_weight, pose_preserve_weight, freeze_betas,
                           per_frame_conf and conf is not None and conf.dim() == 2)
        if smplx:
            cfg.prior_pose_dims = 3 * self.smpl.NUM_BODY_JOINTS       # 63: what the mixture, bending and preserve terms see
            cfg.num_betas_prior = self.smpl.num_betas                 # 10: shape prior / freeze_betas leave the expression alone
        if self.use_lbfgs:
            out = self._fit_lbfgs(cfg, model_idx, tgt, conf, go, bp, be, tr, freeze_betas)
        else:
            out = native.fit_world(self.smpl.native, self.pose_prior.native, cfg, model_idx, tgt, conf, go, bp, be, tr)
        if not run_forward:
            return out, None, None, out["loss"]
        joints, verts = self.final_forward(out, want_vertices=want_vertices)
        return out, joints, verts, out["loss"]

    def chain_supported(self, target_model_indices=None) -> bool:
        """Whether ``fit_chain`` can run this fitter's sequence mode in one call: the Adam branch with kinematic targets (ONE
        launch, ``k2b_fit_sequence``), or the L-BFGS branch on the device driver (``k2b_fit_sequence_lbfgs``: one device-driven
        fit per frame, no host work between the frames); otherwise the caller fits frame by frame."""
        if self.use_lbfgs:
            return getattr(self, "lbfgs_driver", "device") == "device" and (self.smpl_index is not None or target_model_indices is not None)
        idx = self.smpl_index if target_model_indices is None else torch.as_tensor(target_model_indices).reshape(-1).tolist()
        return idx is not None and all(int(i) < self.smpl.num_joints for i in idx)

    def fit_chain(self, init_params: SMPLData, j3d, conf_3d=None, target_model_indices=None,
                  joint_loss_weight: float = 600.0, pose_preserve_weight: float = 5.0, freeze_betas: bool = False,
                  want_vertices: bool = True, run_forward: bool = True):
        """The reference's sequence mode (``api/sequence.py:214-281`` with ``use_previous_frame_init=True``) for ONE
        sequence of T frames in one launch (``k2b_fit_sequence``): frame 0 as ``seq_ind == 0`` from ``init_params``
        (one row), every later frame from its predecessor's result with the preserve term and
        ``num_iters_followup`` iterations.  ``conf_3d``: (K,) or per frame (T, K) as the sequence API passes it.
        Returns ``(params: dict of (T,.) tensors, joints, vertices, per_frame_loss)`` like ``fit_batch``."""
        if not self.chain_supported(target_model_indices):
            raise NotImplementedError("fit_chain: the Adam branch with kinematic targets, or the L-BFGS branch on the device driver")
        per_frame = conf_3d is not None and torch.as_tensor(conf_3d).dim() == 2
        go, bp, be, tr, model_idx, tgt, conf = self._prepare(init_params, j3d, conf_3d, target_model_indices, per_frame,
                                                             num_init=1)
        cfg = self._config(0, joint_loss_weight, pose_preserve_weight, freeze_betas, per_frame)
        cfg.pose_preserve_weight = float(pose_preserve_weight)      # frames >= 1 (frame 0 has no preserve term)
        if self.smpl.packed:
            cfg.prior_pose_dims, cfg.num_betas_prior = 3 * self.smpl.NUM_BODY_JOINTS, self.smpl.num_betas
        T = tgt.shape[0]
        if self.use_lbfgs:
            cfg.freeze_betas = int(bool(freeze_betas))
            out = native.fit_sequence_lbfgs(self.smpl.native, self.pose_prior.native, cfg, int(self.num_iters_first),
                                            int(self.num_iters_followup), model_idx, tgt, conf, go, bp, be, tr, lr=float(self.step_size))
        else:
            out = native.fit_sequence(self.smpl.native, self.pose_prior.native, cfg, int(self.num_iters_followup), model_idx,
                                      tgt.unsqueeze(0), None if conf is None else (conf.unsqueeze(0) if per_frame else conf),
                                      go, bp, be, tr)
            out = {k: v.reshape((T,) + tuple(v.shape[2:])) for k, v in out.items()}
        if not run_forward:
            return out, None, None, out["loss"]
        joints, verts = self.final_forward(out, want_vertices=want_vertices)
        return out, joints, verts, out["loss"]

    def packed_init(self, init):
        """Kernel layout of SMPL-X parameters: ``body_pose`` = all 162 non-root joint values, ``betas`` = betas | expression."""
        B = init.global_orient.shape[0]
        get = lambda name: getattr(init, name, None)
        return SMPLData(global_orient=init.global_orient, transl=init.transl,
                        body_pose=self.smpl.pack_pose(B, body_pose=init.body_pose, jaw_pose=get("jaw_pose"), leye_pose=get("leye_pose"),
                                                      reye_pose=get("reye_pose"), left_hand_pose=get("left_hand_pose"),
                                                      right_hand_pose=get("right_hand_pose")),
                        betas=self.smpl.pack_shape(B, init.betas, get("expression")))

    def fit_params(self, cfg, targets, init, model_idx=None):
        """The hot call alone: ONE fused-fit launch on device tensors that are already in kernel layout
        (`targets` (B,K,3) in the order of this fitter's joint category or of `model_idx`, `init` on the device, packed
        for SMPL-X).  `bench.py` times this + `final_forward`; `fit_batch` is the same two calls behind the reference's
        argument handling."""
        idx = list(self.smpl_index) if model_idx is None else list(model_idx)
        return native.fit_world(self.smpl.native, self.pose_prior.native, cfg, idx, targets, None,
                                init.global_orient, init.body_pose, init.betas, init.transl)

    def final_forward(self, out, want_vertices=True):
        """Final no-grad forward of the reference (world_space.py:258-278): joints (+ vertices) of fitted parameters."""
        return self.smpl.native.lbs(out["global_orient"], out["body_pose"], out["betas"], out["transl"],
                                    want_vertices=want_vertices)

    def _fit_lbfgs(self, cfg, model_idx, tgt, conf, go, bp, be, tr, freeze_betas):
        """LBFGS branch (world_space.py:231-247): ``torch.optim.LBFGS(params, max_iter=num_iters,
        lr=step_size, line_search_fn="strong_wolfe").step(closure)`` per frame, with the closure's
        loss and gradient evaluated by the HIP kernel; final loss re-evaluated afterwards.  One optimiser
        per frame: for B > 1 the reference couples the frames in one line search, this does not (see the module docstring).

        ``self.lbfgs_driver``: "device" (default since round 4: ``k2b_fit_world_lbfgs``, the optimiser itself on the GPU - every
        frame of every batch takes the same path, so a frame's result no longer depends on how a sequence was sharded),
        "host" (the lock-step numpy restatement ``core/lbfgs_batched.py``) or "torch" (``torch.optim.LBFGS`` itself, one frame
        at a time) - the two host drivers are the device path's twins in the tests."""
        max_iter = int(cfg.num_iters)
        B, D = go.shape[0], bp.shape[1]
        NB = be.shape[1]
        preserve = bp.clone()                          # world_space.py:159
        driver = getattr(self, "lbfgs_driver", "device")
        if driver == "device":
            # the optimiser's state machine runs in a kernel of its own, one instance per frame; the call only queues launches
            # (k2b_fit_world_lbfgs: max_eval + 2 rounds of [evaluate-only launch, step launch] + the final loss evaluation)
            cfg.freeze_betas = int(bool(freeze_betas))
            out = native.fit_world_lbfgs(self.smpl.native, self.pose_prior.native, cfg, model_idx, tgt, conf, go, bp, be, tr,
                                         max_iter=max_iter, lr=float(self.step_size), preserve_pose=preserve)
            self.last_lbfgs_rounds = max_iter * 5 // 4 + 2
            return out
        cfg.num_iters, cfg.step_size = 1, 0.0          # evaluate-only launches, driven from the host (the round-2 / round-3 drivers,
        if B > 1 or driver == "host":                  #  kept as the device path's twins: tests, diagnostics)
            return self._fit_lbfgs_lockstep(cfg, max_iter, model_idx, tgt, conf, go, bp, be, tr, preserve, freeze_betas)
        outs = {k: [] for k in ("global_orient", "body_pose", "betas", "transl", "loss")}
        # The optimiser's own arithmetic (two-loop recursion, strong-Wolfe bookkeeping: hundreds of tiny tensor operations per
        # iteration) runs on HOST tensors, as it does in the reference (whose default device is the CPU): on device tensors every
        # one of them is a kernel launch, and a 30-iteration fit took 21 ms of which the evaluate-only launches were 0.3 ms.
        # Per closure call: one upload of the packed parameters, one launch, one download of [gradient | loss].
        dev = self.device
        host = lambda t: t.detach().to("cpu").clone()
        for f in range(B):
            sl = slice(f, f + 1)
            p = [host(go[sl]).requires_grad_(True), host(bp[sl]).requires_grad_(True), host(tr[sl]).requires_grad_(True)]
            beta = host(be[sl])
            # (SMPL-X: the packed shape vector = betas | expression always joins the optimiser; with frozen betas their part of
            #  the gradient is zero (``num_betas_prior``), which leaves them - and L-BFGS's inner products - untouched)
            shape_in_optimiser = not freeze_betas or self.smpl.packed
            if shape_in_optimiser:
                beta.requires_grad_(True)
                p.append(beta)                         # parameter order of world_space.py:215-229
            cf = conf[sl].contiguous() if (conf is not None and conf.dim() == 2) else conf
            pres = preserve[sl].contiguous()
            tgt_f = tgt[sl].contiguous()

            def evaluate(want_grad):
                with torch.no_grad():
                    flat = torch.cat((p[0].detach(), p[1].detach(), beta.detach(), p[2].detach()), dim=1).to(dev)
                    cur = (flat[:, 0:3].contiguous(), flat[:, 3:3 + D].contiguous(), flat[:, 3 + D:3 + D + NB].contiguous(),
                           flat[:, 3 + D + NB:].contiguous())
                    r = native.fit_world(self.smpl.native, self.pose_prior.native, cfg, model_idx, tgt_f, cf, *cur,
                                         preserve_pose=pres, want_grad=want_grad)
                    if not want_grad:
                        return r["loss"], None
                    back = torch.cat((r["grad"], r["loss"][:, None]), dim=1).cpu()
                    return back[:, -1], back[:, :-1]

            def closure():
                loss, g = evaluate(True)
                p[0].grad = g[:, 0:3].clone()
                p[1].grad = g[:, 3:3 + D].clone()
                p[2].grad = g[:, 3 + D + NB:].clone()
                if shape_in_optimiser:
                    beta.grad = g[:, 3 + D:3 + D + NB].clone()
                return loss.sum()

            torch.optim.LBFGS(p, max_iter=max_iter, lr=float(self.step_size), line_search_fn="strong_wolfe").step(closure)
            final_loss, _ = evaluate(False)            # world_space.py:245-246
            outs["global_orient"].append(p[0].detach().to(dev)); outs["body_pose"].append(p[1].detach().to(dev))
            outs["transl"].append(p[2].detach().to(dev)); outs["betas"].append(beta.detach().to(dev)); outs["loss"].append(final_loss)
        return {k: torch.cat(v, dim=0).contiguous() for k, v in outs.items()}

    def _fit_lbfgs_lockstep(self, cfg, max_iter, model_idx, tgt, conf, go, bp, be, tr, preserve, freeze_betas):
        """B > 1 frames in the L-BFGS branch: the B per-frame optimisers advance in lock-step (``core/lbfgs_batched.py``, torch's
        L-BFGS / strong-Wolfe restated and vectorised over the frames), so ONE evaluate-only launch over all B frames serves
        the pending closure call of every frame - ~max_iter * 5 / 4 launches for the whole batch instead of that many per
        frame, and the optimiser's bookkeeping is a few numpy operations per round instead of B x hundreds of tiny tensor
        operations.  Frames stay independent (each has its own history, line search and stopping rule).  Per round: one
        upload of the (B, P) points, one launch, one download of [gradient | loss]."""
        from ..lbfgs_batched import BatchedLBFGS
        B, D, NB = go.shape[0], bp.shape[1], be.shape[1]
        dev = self.device
        shape_in_optimiser = not freeze_betas or self.smpl.packed          # as in the per-frame path below
        start = torch.cat((go, bp, be, tr), dim=1)                         # kernel layout [go | pose | shape | transl]
        free = np.ones(3 + D + NB + 3, dtype=bool)
        if not shape_in_optimiser:
            free[3 + D: 3 + D + NB] = False
        free_t = torch.as_tensor(free, device=dev)
        flat = start.clone()

        def launch(x_free, want_grad):
            with torch.no_grad():
                flat[:, free_t] = torch.from_numpy(np.ascontiguousarray(x_free, dtype=np.float32)).to(dev)
                cur = (flat[:, 0:3].contiguous(), flat[:, 3:3 + D].contiguous(), flat[:, 3 + D:3 + D + NB].contiguous(),
                       flat[:, 3 + D + NB:].contiguous())
                return native.fit_world(self.smpl.native, self.pose_prior.native, cfg, model_idx, tgt, conf, *cur,
                                        preserve_pose=preserve, want_grad=want_grad), cur

        def evaluate(x_free):
            r, _ = launch(x_free, True)
            back = torch.cat((r["grad"], r["loss"][:, None]), dim=1).cpu().numpy()
            return back[:, -1].astype(np.float64), back[:, :-1][:, free]

        opt = BatchedLBFGS(evaluate, start.cpu().numpy()[:, free], lr=float(self.step_size), max_iter=max_iter)
        x = opt.run()
        r, cur = launch(x, False)                                          # final loss re-evaluated (world_space.py:245-246)
        self.last_lbfgs_rounds = opt.rounds
        return {"global_orient": cur[0], "body_pose": cur[1], "betas": cur[2], "transl": cur[3], "loss": r["loss"]}

    def fit_frame(self, init_params: SMPLData, j3d: torch.Tensor, conf_3d: Optional[torch.Tensor] = None,
                  seq_ind: int = 0, target_model_indices: Optional[torch.Tensor] = None,
                  joint_loss_weight: float = 600.0, pose_preserve_weight: float = 5.0,
                  freeze_betas: bool = False) -> BodyModelFitResult:
        """Fit one frame (or a batch) with the reference's ``fit_frame`` contract:
        inputs are not modified, outputs are detached tensors, ``loss`` is the batch sum of
        the last iteration's loss evaluated before its step (world_space.py:256)."""
        out, joints, verts, loss = self.fit_batch(
            init_params, j3d, conf_3d, seq_ind, target_model_indices, joint_loss_weight, pose_preserve_weight,
            freeze_betas, per_frame_conf=False)
        return BodyModelFitResult(params=self.result_params(out, init_params), vertices=verts, joints=joints, loss=loss.sum())

    def result_params(self, out, init_params, rows: Optional[slice] = None):
        """Fitted parameters as the data class the reference returns for this model / input type; `rows` selects
        frames of a batched result (hand / face fields a 24-joint fit does not touch are carried from `init_params`)."""
        if rows is not None:
            out = {k: v[rows] for k, v in out.items() if k != "loss"}
        if self.smpl.model_type == "smplh" and self.smpl.packed:
            u = self.smpl.unpack(out["body_pose"], out["betas"])
            return SMPLHData(betas=u["betas"], global_orient=out["global_orient"], body_pose=u["body_pose"], transl=out["transl"],
                             left_hand_pose=u["left_hand_pose"], right_hand_pose=u["right_hand_pose"])
        if self.smpl.model_type == "smplx":
            u = self.smpl.unpack(out["body_pose"], out["betas"])
            fitted = SMPLXData(betas=u["betas"], global_orient=out["global_orient"], body_pose=u["body_pose"],
                               transl=out["transl"], left_hand_pose=u["left_hand_pose"], right_hand_pose=u["right_hand_pose"],
                               expression=u["expression"], jaw_pose=u["jaw_pose"], leye_pose=u["leye_pose"],
                               reye_pose=u["reye_pose"])
            return fitted
        fields = dict(betas=out["betas"], global_orient=out["global_orient"], body_pose=out["body_pose"],
                      transl=out["transl"])
        if isinstance(init_params, SMPLXData):
            keep = ("left_hand_pose", "right_hand_pose", "expression", "jaw_pose", "leye_pose", "reye_pose")
            fitted = SMPLXData(**fields, **{k: _detached(getattr(init_params, k)) for k in keep})
        elif isinstance(init_params, SMPLHData):
            keep = ("left_hand_pose", "right_hand_pose")
            fitted = SMPLHData(**fields, **{k: _detached(getattr(init_params, k)) for k in keep})
        else:
            fitted = SMPLData(**fields)
        return fitted


def _detached(x):
    return x.detach() if isinstance(x, torch.Tensor) else x
