"""ORACLE TOOLING: generate golden vectors by running the REAL reference.

Runs only in the build container (``/root/reference`` does not exist on the GPU
box).  It imports the reference's own ``WorldSpaceFitter`` /
``guess_init_transl_from_root`` / ``MaxMixturePrior`` (reference
``keypoints2body/core/fitters/world_space.py``, ``core/prior.py``, ``core/losses.py``)
and drives them with

* the synthetic SMPL-shaped model of ``keypoints2body_amd.synthetic`` wrapped in
  ``oracle.smpl_torch.TorchSMPL`` as the ``model=`` body-model plugin (the
  reference's own forward lives in the absent third-party ``smplx``), and
* a synthetic ``gmm_08.pkl`` written by this script (our own file) into a
  scratch ``./data/models`` that the reference reads relative to the CWD
  (reference ``core/fitters/world_space.py:87-91``).

Nothing of the reference's source is copied: the package root is registered as
an empty module whose ``__path__`` points at ``/root/reference/keypoints2body`` so
that its ``__init__`` (which needs the absent ``h5py``/``smplx``) is not executed.

Outputs: ``tests/golden/world_fit_*.npz`` (inputs + reference outputs) and
``tests/golden/gmm_synth.npz`` (mixture + the buffers the reference derived).

Usage:  python oracle/gen_golden.py [--only SUBSTRING]
(``--only`` still runs every case - they share seeded state - but writes only the files whose name contains
SUBSTRING, so that adding a case does not rewrite the zip timestamps of the other fixtures.)
"""
from __future__ import annotations

import os
import pickle
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))

from keypoints2body_amd import synthetic  # noqa: E402
from oracle.smpl_torch import TorchSMPL  # noqa: E402

REF_ROOT = Path("/root/reference/keypoints2body")
GOLDEN = REPO / "tests" / "golden"
TRACE_ITERS = (1, 2, 10, 50, 100)
N_SAMPLED_VERTS = 256


def import_reference():
    """Register stub parents so reference sub-modules import with torch alone."""
    for name, sub in (("keypoints2body", ""), ("keypoints2body.core", "core")):
        mod = types.ModuleType(name)
        mod.__path__ = [str(REF_ROOT / sub)]
        sys.modules[name] = mod
    from keypoints2body.core.fitters.world_space import (  # type: ignore
        WorldSpaceFitter, guess_init_transl_from_root)
    from keypoints2body.core.prior import MaxMixturePrior  # type: ignore
    from keypoints2body.models.smpl_data import SMPLData  # type: ignore
    return WorldSpaceFitter, guess_init_transl_from_root, MaxMixturePrior, SMPLData


def sample_vertex_ids(V: int) -> np.ndarray:
    return ((97 + 26 * np.arange(N_SAMPLED_VERTS)) % V).astype(np.int64)


class Recorder:
    """Wraps the model plugin to snapshot the leaf parameters at every forward call.

    The reference's Adam loop calls the model once per iteration *before* the step
    (world_space.py:251-255), so call n sees the parameters after n-1 steps and the
    final no-grad forward (world_space.py:258-278) sees them after all steps.
    """

    def __init__(self, model):
        self.model = model
        self.snaps = []

    def __call__(self, **kw):
        self.snaps.append({k: kw[k].detach().clone() for k in ("global_orient", "body_pose", "betas", "transl")
                           if k in kw and kw[k] is not None})
        return self.model(**kw)

    def __getattr__(self, name):
        return getattr(self.model, name)


def run_case(name, ref, model, *, init, j3d, conf, seq_ind, num_iters, category,
             target_model_indices=None, freeze_betas=False, per_frame_calls=True, extra=None):
    WorldSpaceFitter, _, _, SMPLData = ref
    B = j3d.shape[0]
    init = {k: v[:B] for k, v in init.items()}
    iters_first = num_iters if seq_ind == 0 else 7
    iters_follow = num_iters if seq_ind > 0 else 7
    rec = Recorder(model)
    fitter = WorldSpaceFitter(rec, step_size=1e-2, num_iters_first=iters_first,
                              num_iters_followup=iters_follow, use_lbfgs=False,
                              joints_category=category, device=torch.device("cpu"))
    K = j3d.shape[1]
    trace_iters = [t for t in TRACE_ITERS if t <= num_iters]
    out = {k: [] for k in ("go", "bp", "be", "tr", "joints", "verts_sampled", "verts_sum", "loss")}
    tr = {k: [] for k in ("go", "bp", "be", "tr")}
    all_losses = []
    groups = [slice(i, i + 1) for i in range(B)] if per_frame_calls else [slice(0, B)]
    for g in groups:
        rec.snaps.clear()
        losses = []
        # per-iteration loss: the reference only returns the last one, so capture
        # each iteration's loss by wrapping Tensor.backward on this call path.
        orig_backward = torch.Tensor.backward

        def spy_backward(self, *a, **k):
            losses.append(float(self.detach()))
            return orig_backward(self, *a, **k)

        torch.Tensor.backward = spy_backward
        try:
            res = fitter.fit_frame(
                init_params=SMPLData(betas=init["betas"][g], global_orient=init["global_orient"][g],
                                     body_pose=init["body_pose"][g], transl=init["transl"][g]),
                j3d=j3d[g], conf_3d=conf, seq_ind=seq_ind,
                target_model_indices=target_model_indices,
                joint_loss_weight=600.0, pose_preserve_weight=5.0, freeze_betas=freeze_betas)
        finally:
            torch.Tensor.backward = orig_backward
        assert len(rec.snaps) == num_iters + 1 and len(losses) == num_iters
        all_losses.append(losses)
        for t in trace_iters:          # params after t steps = snapshot index t
            s = rec.snaps[t]
            tr["go"].append(s["global_orient"]); tr["bp"].append(s["body_pose"])
            tr["be"].append(s["betas"]); tr["tr"].append(s["transl"])
        p = res.params
        out["go"].append(p.global_orient); out["bp"].append(p.body_pose)
        out["be"].append(p.betas); out["tr"].append(p.transl)
        out["joints"].append(res.joints)
        vid = sample_vertex_ids(res.vertices.shape[1])
        out["verts_sampled"].append(res.vertices[:, vid])
        out["verts_sum"].append(res.vertices.double().sum(dim=1))
        out["loss"].append(res.loss.reshape(1))

    nG = len(groups)
    cat = lambda xs: torch.cat(xs, dim=0).numpy()

    def stack_trace(xs):  # list ordered [group][iter] -> (n_trace, B, D)
        per_iter = [torch.cat([xs[gi * len(trace_iters) + ti] for gi in range(nG)], dim=0)
                    for ti in range(len(trace_iters))]
        return torch.stack(per_iter, dim=0).numpy()

    payload = dict(
        case=name, category=category, seq_ind=seq_ind, num_iters=num_iters,
        freeze_betas=int(freeze_betas), per_frame_calls=int(per_frame_calls),
        model_fingerprint=np.uint64(model_fingerprint),
        init_global_orient=init["global_orient"].numpy(), init_body_pose=init["body_pose"].numpy(),
        init_betas=init["betas"].numpy(), init_transl=init["transl"].numpy(),
        j3d=j3d.numpy(), conf=(conf.numpy() if conf is not None else np.zeros(0, np.float32)),
        has_conf=int(conf is not None),
        target_model_indices=(target_model_indices.numpy() if target_model_indices is not None
                              else np.zeros(0, np.int64)),
        trace_iters=np.array(trace_iters),
        trace_global_orient=stack_trace(tr["go"]), trace_body_pose=stack_trace(tr["bp"]),
        trace_betas=stack_trace(tr["be"]), trace_transl=stack_trace(tr["tr"]),
        # losses[g][i] = scalar the reference back-propagated at iteration i+1 (sum over the call's batch)
        iter_losses=np.array(all_losses, dtype=np.float64),
        out_global_orient=cat(out["go"]), out_body_pose=cat(out["bp"]), out_betas=cat(out["be"]),
        out_transl=cat(out["tr"]), out_joints=cat(out["joints"]),
        out_verts_sampled=cat(out["verts_sampled"]), out_verts_sum=cat(out["verts_sum"]),
        sampled_vertex_ids=sample_vertex_ids(model.v_template.shape[0]),
        out_loss=cat(out["loss"]),
    )
    if extra:
        payload.update(extra)
    np.savez_compressed(GOLDEN / f"world_fit_{name}.npz", **payload)
    print(f"[golden] {name}: B={B} iters={num_iters} final loss(es)={payload['out_loss']}")


def main():
    global model_fingerprint
    if "--only" in sys.argv:
        only, real_save = sys.argv[sys.argv.index("--only") + 1], np.savez_compressed

        def filtered_save(path, **kw):
            if only in Path(path).name:
                real_save(path, **kw)
            else:
                print(f"[golden] (not written: {Path(path).name})")
        np.savez_compressed = filtered_save
    torch.manual_seed(0)
    torch.set_num_threads(8)
    GOLDEN.mkdir(parents=True, exist_ok=True)
    consts = synthetic.make_body_model(seed=0)
    model_fingerprint = consts.fingerprint()
    model = TorchSMPL(consts)
    gmm = synthetic.make_gmm(seed=0)

    scratch = tempfile.mkdtemp(prefix="k2b_golden_")
    os.makedirs(os.path.join(scratch, "data", "models"))
    with open(os.path.join(scratch, "data", "models", "gmm_08.pkl"), "wb") as f:
        pickle.dump({"means": gmm.means, "covars": gmm.covars, "weights": gmm.weights}, f)
    os.chdir(scratch)

    ref = import_reference()
    _, guess_transl, MaxMixturePrior, _ = ref

    # The buffers the reference derives from the mixture (prior.py:133-163).
    prior = MaxMixturePrior(prior_folder="./data/models/", num_gaussians=8, dtype=torch.float32)
    np.savez_compressed(
        GOLDEN / "gmm_synth.npz",
        means=gmm.means, covars=gmm.covars.astype(np.float32), weights=gmm.weights,
        ref_means=prior.means.numpy(), ref_precisions=prior.precisions.numpy(),
        ref_nll_weights=prior.nll_weights.numpy(),
    )
    # known-answer values of the prior itself on a few poses
    probe_pose = torch.tensor((0.2 * synthetic.normalish(40, (5, 69), 0)).astype(np.float32))
    np.savez_compressed(GOLDEN / "prior_probe.npz", pose=probe_pose.numpy(),
                        value=prior(probe_pose, None).numpy())

    T = 6
    poses = synthetic.make_poses(T, seed=0)
    tt = lambda a: torch.tensor(np.asarray(a))
    with torch.no_grad():
        gt = model(global_orient=tt(poses.global_orient), body_pose=tt(poses.body_pose),
                   betas=tt(poses.betas), transl=tt(poses.transl))
    j22 = gt.joints[:, :22].clone()
    j24 = gt.joints[:, :24].clone()
    noisy22 = j22 + tt(synthetic.target_noise(T, 22, seed=0))

    def zero_init(j3d, category):
        pose = torch.zeros(j3d.shape[0], 72)
        betas = torch.zeros(j3d.shape[0], 10)
        transl = guess_transl(model, pose, betas, j3d, joints_category=category)
        return dict(global_orient=pose[:, :3].clone(), body_pose=pose[:, 3:].clone(), betas=betas, transl=transl)

    # 1. headline: six independent frames, zero init, 100 Adam iterations, AMASS-22
    init = zero_init(j22, "AMASS")
    run_case("amass_zero_init", ref, model, init=init, j3d=j22, conf=None, seq_ind=0,
             num_iters=100, category="AMASS")

    # 2. same inputs, one batched call (B=6) -> reference's own batch semantics
    run_case("amass_batched", ref, model, init=init, j3d=j22, conf=None, seq_ind=0,
             num_iters=100, category="AMASS", per_frame_calls=False)

    # 3. noisy targets, non-trivial "mean pose" init, confidences (fix_foot style 1.5 + some 0.7)
    mean_pose = tt((0.1 * synthetic.normalish(41, (1, 72), 0)).astype(np.float32)).repeat(T, 1)
    mean_shape = tt((0.2 * synthetic.normalish(42, (1, 10), 0)).astype(np.float32)).repeat(T, 1)
    conf = torch.ones(22); conf[[7, 8, 10, 11]] = 1.5; conf[[15, 20]] = 0.7
    init3 = dict(global_orient=mean_pose[:, :3].clone(), body_pose=mean_pose[:, 3:].clone(), betas=mean_shape,
                 transl=guess_transl(model, mean_pose, mean_shape, noisy22, joints_category="AMASS"))
    run_case("amass_noisy_conf", ref, model, init=init3, j3d=noisy22, conf=conf, seq_ind=0,
             num_iters=100, category="AMASS")

    # 4. follow-up frame semantics: seq_ind>0 (preserve term on), 50 iterations, warm start
    warm = dict(global_orient=tt(poses.global_orient) + 0.05, body_pose=tt(poses.body_pose) * 0.8,
                betas=tt(poses.betas) * 0.5, transl=tt(poses.transl) + 0.02)
    run_case("amass_followup", ref, model, init=warm, j3d=noisy22, conf=conf, seq_ind=3,
             num_iters=50, category="AMASS")

    # 5. frozen betas
    run_case("amass_freeze_betas", ref, model, init=init3, j3d=noisy22[:3], conf=None, seq_ind=0,
             num_iters=30, category="AMASS", freeze_betas=True)

    # 6. SMPL24 category (all 24 kinematic joints observed)
    init6 = zero_init(j24, "SMPL24")
    run_case("smpl24_zero_init", ref, model, init=init6, j3d=j24[:3], conf=None, seq_ind=0,
             num_iters=100, category="SMPL24")

    # 7. GENERIC category with explicit target_model_indices: a shuffled subset of kinematic joints
    rest0 = model(global_orient=torch.zeros(T, 3), body_pose=torch.zeros(T, 69), betas=torch.zeros(T, 10)).joints
    idx = torch.tensor([0, 2, 1, 5, 4, 8, 7, 12, 15, 17, 16, 19, 18, 21, 20, 23, 22], dtype=torch.long)
    jgen = gt.joints[:, idx].clone()
    cgen = torch.ones(idx.numel()); cgen[[3, 9]] = 0.5
    init7 = dict(global_orient=torch.zeros(T, 3), body_pose=torch.zeros(T, 69), betas=torch.zeros(T, 10),
                 transl=(jgen[:, 0] - rest0[:, 0]).detach())
    run_case("generic_indices", ref, model, init=init7, j3d=jgen[:3], conf=cgen, seq_ind=0,
             num_iters=40, category="GENERIC", target_model_indices=idx)

    # 8. GENERIC with vertex-selected joints (24..44) in the loss: gradient flows through LBS vertices
    idx8 = torch.tensor(list(range(22)) + [24, 25, 30, 37, 44], dtype=torch.long)
    jgen8 = gt.joints[:, idx8].clone()
    run_case("generic_vertex_joints", ref, model, init=init7, j3d=jgen8[:2], conf=torch.ones(27), seq_ind=0,
             num_iters=30, category="GENERIC", target_model_indices=idx8)

    # ---- camera-space two-stage fitter (reference core/fitters/camera_space.py) -----------------------
    from keypoints2body.core.fitters.camera_space import CameraSpaceFitter  # type: ignore
    _, _, _, SMPLData = ref
    # The reference's default start (camera_t = mean torso offset) makes d loss / d camera_t at the first
    # stage-1 step rounding noise around zero, and Adam's first step is scale-free, so the default-start
    # result is only defined to ~1e-3 (it differs by that much between two CPUs running this script).
    # The pinned cases therefore use the fitter's own `init_cam_t` argument with a start 2-5 cm off
    # the mean offset; "default_start" records the default behaviour for a looser check.
    from keypoints2body.core.fitters.camera_space import guess_init_3d  # type: ignore
    off = torch.tensor([[0.03, -0.02, 0.05]])
    for name, iters, seq_ind, freeze, cf, src, use_off in (("full", 50, 0, False, conf, noisy22, True),
                                                           ("followup_frozen", 20, 2, True, conf, noisy22, True),
                                                           ("default_start", 50, 0, False, conf, noisy22, False)):
        fitter = CameraSpaceFitter(model, step_size=1e-2, num_iters=iters, use_lbfgs=False,
                                   joints_category="AMASS", device=torch.device("cpu"))
        outs = {k: [] for k in ("go", "bp", "be", "tr", "joints", "loss", "verts_sampled", "t0")}
        n = 4
        for i in range(n):
            sl = slice(i, i + 1)
            t0 = None
            if use_off:
                with torch.no_grad():
                    j0 = model(global_orient=init3["global_orient"][sl], body_pose=init3["body_pose"][sl],
                               betas=init3["betas"][sl]).joints
                t0 = guess_init_3d(j0, src[sl], "AMASS") + off
            outs["t0"].append(t0 if t0 is not None else torch.zeros(1, 3))
            res = fitter.fit_frame(SMPLData(betas=init3["betas"][sl], global_orient=init3["global_orient"][sl],
                                            body_pose=init3["body_pose"][sl]),
                                   src[sl], conf_3d=cf, seq_ind=seq_ind, joint_loss_weight=600.0,
                                   pose_preserve_weight=5.0, freeze_betas=freeze, init_cam_t=t0)
            p = res.params
            outs["go"].append(p.global_orient); outs["bp"].append(p.body_pose); outs["be"].append(p.betas)
            outs["tr"].append(p.transl); outs["joints"].append(res.joints); outs["loss"].append(res.loss.reshape(1))
            outs["verts_sampled"].append(res.vertices[:, sample_vertex_ids(res.vertices.shape[1])])
        cat = lambda xs: torch.cat(xs, dim=0).detach().numpy()
        np.savez_compressed(
            GOLDEN / f"camera_fit_{name}.npz", case=name, num_iters=iters, seq_ind=seq_ind, freeze_betas=int(freeze),
            has_conf=int(cf is not None), conf=(cf.numpy() if cf is not None else np.zeros(0, np.float32)),
            model_fingerprint=np.uint64(model_fingerprint),
            init_global_orient=init3["global_orient"][:n].numpy(), init_body_pose=init3["body_pose"][:n].numpy(),
            init_betas=init3["betas"][:n].numpy(), j3d=src[:n].numpy(),
            has_init_cam_t=int(use_off), init_cam_t=cat(outs["t0"]),
            out_global_orient=cat(outs["go"]), out_body_pose=cat(outs["bp"]), out_betas=cat(outs["be"]),
            out_transl=cat(outs["tr"]), out_joints=cat(outs["joints"]), out_loss=cat(outs["loss"]),
            out_verts_sampled=cat(outs["verts_sampled"]), sampled_vertex_ids=sample_vertex_ids(6890))
        print(f"[golden] camera {name}: iters={iters} losses={cat(outs['loss'])}")

    # ---- camera fitter, GENERIC targets with vertex-selected joints (model index >= 24) in both stages --------
    fitter = CameraSpaceFitter(model, step_size=1e-2, num_iters=30, use_lbfgs=False,
                               joints_category="GENERIC", device=torch.device("cpu"))
    outs = {k: [] for k in ("go", "bp", "be", "tr", "joints", "loss", "verts_sampled", "t0")}
    n = 2
    conf8 = torch.ones(27); conf8[[4, 23]] = 0.6
    for i in range(n):
        sl = slice(i, i + 1)
        with torch.no_grad():
            j0 = model(global_orient=init7["global_orient"][sl], body_pose=init7["body_pose"][sl], betas=init7["betas"][sl]).joints
        t0 = (jgen8[sl, 0] - j0[:, int(idx8[0])]) + off
        outs["t0"].append(t0)
        res = fitter.fit_frame(SMPLData(betas=init7["betas"][sl], global_orient=init7["global_orient"][sl],
                                        body_pose=init7["body_pose"][sl]),
                               jgen8[sl], conf_3d=conf8, seq_ind=0, target_model_indices=idx8, joint_loss_weight=600.0,
                               pose_preserve_weight=5.0, freeze_betas=False, init_cam_t=t0)
        p = res.params
        outs["go"].append(p.global_orient); outs["bp"].append(p.body_pose); outs["be"].append(p.betas)
        outs["tr"].append(p.transl); outs["joints"].append(res.joints); outs["loss"].append(res.loss.reshape(1))
        outs["verts_sampled"].append(res.vertices[:, sample_vertex_ids(res.vertices.shape[1])])
    cat = lambda xs: torch.cat(xs, dim=0).detach().numpy()
    np.savez_compressed(
        GOLDEN / "camera_fit_generic_vertex_joints.npz", case="generic_vertex_joints", num_iters=30, seq_ind=0, freeze_betas=0,
        has_conf=1, conf=conf8.numpy(), model_fingerprint=np.uint64(model_fingerprint), target_model_indices=idx8.numpy(),
        init_global_orient=init7["global_orient"][:n].numpy(), init_body_pose=init7["body_pose"][:n].numpy(),
        init_betas=init7["betas"][:n].numpy(), j3d=jgen8[:n].numpy(), has_init_cam_t=1, init_cam_t=cat(outs["t0"]),
        out_global_orient=cat(outs["go"]), out_body_pose=cat(outs["bp"]), out_betas=cat(outs["be"]),
        out_transl=cat(outs["tr"]), out_joints=cat(outs["joints"]), out_loss=cat(outs["loss"]),
        out_verts_sampled=cat(outs["verts_sampled"]), sampled_vertex_ids=sample_vertex_ids(6890))
    print(f"[golden] camera generic_vertex_joints: losses={cat(outs['loss'])}")

    # ---- LBFGS branch of the camera-space fitter ---------------------------------------------------------
    for name, iters, seq_ind, freeze in (("first", 20, 0, False), ("followup_frozen", 10, 3, True)):
        fitter = CameraSpaceFitter(model, step_size=1e-2, num_iters=iters, use_lbfgs=True,
                                   joints_category="AMASS", device=torch.device("cpu"))
        outs = {k: [] for k in ("go", "bp", "be", "tr", "joints", "loss", "t0")}
        n = 3
        gen = torch.Generator().manual_seed(4321)
        pert, pdev, jdev = (np.zeros((n, 10), np.float32) for _ in range(3))
        for i in range(n):
            sl = slice(i, i + 1)
            with torch.no_grad():
                j0 = model(global_orient=init3["global_orient"][sl], body_pose=init3["body_pose"][sl],
                           betas=init3["betas"][sl]).joints
            t0 = guess_init_3d(j0, noisy22[sl], "AMASS") + off
            outs["t0"].append(t0)
            call = lambda go, bp, t: fitter.fit_frame(
                SMPLData(betas=init3["betas"][sl], global_orient=go, body_pose=bp), noisy22[sl], conf_3d=conf,
                seq_ind=seq_ind, joint_loss_weight=600.0, pose_preserve_weight=5.0, freeze_betas=freeze, init_cam_t=t)
            res = call(init3["global_orient"][sl], init3["body_pose"][sl], t0)
            p = res.params
            outs["go"].append(p.global_orient); outs["bp"].append(p.body_pose); outs["be"].append(p.betas)
            outs["tr"].append(p.transl); outs["joints"].append(res.joints); outs["loss"].append(res.loss.reshape(1))
            jerr0 = (res.joints[:, :22] + p.transl[:, None] - noisy22[sl]).norm(dim=-1).mean()
            for trial in range(10):        # the reference's own spread under rounding-level perturbations (see the world LBFGS cases)
                nz = lambda x: x * (1 + 2e-6 * torch.randn(x.shape, generator=gen))
                rp = call(nz(init3["global_orient"][sl]), nz(init3["body_pose"][sl]), nz(t0))
                pert[i, trial] = float(rp.loss)
                pdev[i, trial] = max(float((getattr(rp.params, k) - getattr(p, k)).abs().max())
                                     for k in ("global_orient", "body_pose", "betas", "transl"))
                jdev[i, trial] = abs(float((rp.joints[:, :22] + rp.params.transl[:, None] - noisy22[sl]).norm(dim=-1).mean() - jerr0))
        cat = lambda xs: torch.cat(xs, dim=0).detach().numpy()
        np.savez_compressed(
            GOLDEN / f"lbfgs_camera_{name}.npz", case=name, seq_ind=seq_ind, max_iter=iters, freeze_betas=int(freeze),
            conf=conf.numpy(), j3d=noisy22[:n].numpy(), init_cam_t=cat(outs["t0"]),
            init_global_orient=init3["global_orient"][:n].numpy(), init_body_pose=init3["body_pose"][:n].numpy(),
            init_betas=init3["betas"][:n].numpy(),
            out_global_orient=cat(outs["go"]), out_body_pose=cat(outs["bp"]), out_betas=cat(outs["be"]),
            out_transl=cat(outs["tr"]), out_loss=cat(outs["loss"]), out_joints=cat(outs["joints"]), out_loss_perturbed=pert,
            out_param_dev_perturbed=pdev, out_joint_err_dev_perturbed=jdev)
        print(f"[golden] camera lbfgs {name}: losses={cat(outs['loss'])} spread={pert.min(1)}..{pert.max(1)}")

    # ---- LBFGS branch of the world fitter (the reference's default) --------------------------------------
    WorldSpaceFitter = ref[0]
    for name, seq_ind, iters, freeze in (("first", 0, 30, False), ("followup", 4, 10, False), ("frozen", 0, 15, True)):
        fitter = WorldSpaceFitter(model, step_size=1e-2, num_iters_first=iters, num_iters_followup=iters,
                                  use_lbfgs=True, joints_category="AMASS", device=torch.device("cpu"))
        src_init = init3 if seq_ind == 0 else warm
        outs = {k: [] for k in ("go", "bp", "be", "tr", "loss", "joints")}
        n = 3
        for i in range(n):
            sl = slice(i, i + 1)
            res = fitter.fit_frame(SMPLData(betas=src_init["betas"][sl], global_orient=src_init["global_orient"][sl],
                                            body_pose=src_init["body_pose"][sl], transl=src_init["transl"][sl]),
                                   noisy22[sl], conf_3d=conf, seq_ind=seq_ind, joint_loss_weight=600.0,
                                   pose_preserve_weight=5.0, freeze_betas=freeze)
            p = res.params
            outs["go"].append(p.global_orient); outs["bp"].append(p.body_pose); outs["be"].append(p.betas)
            outs["tr"].append(p.transl); outs["loss"].append(res.loss.reshape(1)); outs["joints"].append(res.joints)
        # The LBFGS branch is chaotic at this iteration count: its strong-Wolfe line search branches on
        # rounding.  Record the reference's OWN spread: the same calls with the initial parameters
        # perturbed by 2e-6 relative (a few fp32 ulps), so that tests can gate on that envelope.
        gen = torch.Generator().manual_seed(1234)
        pert = np.zeros((n, 10), np.float32)
        for i in range(n):
            sl = slice(i, i + 1)
            for trial in range(10):
                nz = lambda x: x * (1 + 2e-6 * torch.randn(x.shape, generator=gen))
                res = fitter.fit_frame(SMPLData(betas=src_init["betas"][sl], global_orient=nz(src_init["global_orient"][sl]),
                                                body_pose=nz(src_init["body_pose"][sl]), transl=nz(src_init["transl"][sl])),
                                       noisy22[sl], conf_3d=conf, seq_ind=seq_ind, joint_loss_weight=600.0,
                                       pose_preserve_weight=5.0, freeze_betas=freeze)
                pert[i, trial] = float(res.loss)
        cat = lambda xs: torch.cat(xs, dim=0).detach().numpy()
        np.savez_compressed(
            GOLDEN / f"lbfgs_world_{name}.npz", case=name, seq_ind=seq_ind, max_iter=iters, freeze_betas=int(freeze),
            out_loss_perturbed=pert,
            conf=conf.numpy(), j3d=noisy22[:n].numpy(),
            init_global_orient=src_init["global_orient"][:n].numpy(), init_body_pose=src_init["body_pose"][:n].numpy(),
            init_betas=src_init["betas"][:n].numpy(), init_transl=src_init["transl"][:n].numpy(),
            out_global_orient=cat(outs["go"]), out_body_pose=cat(outs["bp"]), out_betas=cat(outs["be"]),
            out_transl=cat(outs["tr"]), out_loss=cat(outs["loss"]), out_joints=cat(outs["joints"]))
        print(f"[golden] lbfgs {name}: losses={cat(outs['loss'])}")

    # ---- multi-frame shape pre-pass (reference core/shape.py, LBFGS branch) ------------------------------
    from keypoints2body.core.shape import optimize_shape_multi_frame  # type: ignore
    Tn = 6
    # a sequence the pass can explain: every frame at the mean pose, one common body shape, own translation
    with torch.no_grad():
        true_betas = tt(poses.betas[:1]).repeat(Tn, 1)
        seq = model(global_orient=mean_pose[:Tn, :3], body_pose=mean_pose[:Tn, 3:], betas=true_betas,
                    transl=tt(poses.transl[:Tn])).joints[:, :22] + tt(synthetic.target_noise(Tn, 22, seed=5, scale=0.002))
    betas_opt = optimize_shape_multi_frame(model, init_betas=mean_shape[:1], pose_init=mean_pose[:Tn], j3d_world=seq,
                                           joints_category="AMASS", num_iters=40, step_size=1e-1, use_lbfgs=True,
                                           device=torch.device("cpu"), frame_indices=list(range(4)),
                                           joints3d_conf=conf, shape_prior_weight=5.0)
    np.savez_compressed(GOLDEN / "shape_pass.npz", j3d=seq.numpy(), conf=conf.numpy(), mean_pose=mean_pose[:1].numpy(),
                        init_betas=mean_shape[:1].numpy(), num_shape_frames=4, num_shape_iters=40,
                        true_betas=true_betas[:1].numpy(), out_betas=betas_opt.numpy())
    print("[golden] shape pass betas:", betas_opt.numpy().round(4))

    print("golden fixtures written to", GOLDEN)


if __name__ == "__main__":
    main()
