"""ORACLE TOOLING: SMPL-X goldens by running the REAL reference fitter (build container only).

The reference's ``WorldSpaceFitter.fit_frame`` is driven with ``SMPLXData`` initial parameters
(``core/fitters/world_space.py:126-151,173-192,215-229``: hands, expression, jaw and eye poses join the Adam parameter list and
are passed to the model) and the oracle's ``TorchSMPLX`` as the ``model=`` plugin (55-joint tree, full hand poses).  One
thing has to be defined by this engine: the reference hands the 63-D SMPL-X body pose to a 69-D mixture, which raises
(``core/prior.py:183``, SURVEY.md note N3).  Here the reference's OWN prior object is wrapped so that it is evaluated at
``[body_pose | 0 x 6]``; every other line that runs is the reference's.  PARITY UNPINNED at the smplx boundary, as for SMPL.

Outputs: ``tests/golden/smplx_fit_*.npz``.   Usage:  python oracle/gen_golden_smplx.py
"""
from __future__ import annotations

import os
import pickle
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))

from keypoints2body_amd import synthetic  # noqa: E402
from oracle.gen_golden import GOLDEN, Recorder, import_reference, sample_vertex_ids  # noqa: E402
from oracle.smpl_torch import TorchSMPLX  # noqa: E402

FIELDS = ("global_orient", "body_pose", "transl", "left_hand_pose", "right_hand_pose", "expression", "jaw_pose",
          "leye_pose", "reye_pose", "betas")
TRACE_ITERS = (1, 2, 10, 50, 100)


class PaddedPrior(torch.nn.Module):
    """The reference's MaxMixturePrior evaluated at [pose | 0 ...] (the only line of the SMPL-X path that is ours)."""

    def __init__(self, ref_prior):
        super().__init__()
        self.ref = ref_prior

    def forward(self, pose, betas):
        d = self.ref.means.shape[1]
        if pose.shape[1] < d:
            pose = torch.nn.functional.pad(pose, (0, d - pose.shape[1]))
        return self.ref(pose, betas)


class RecorderX(Recorder):
    def __call__(self, **kw):
        self.snaps.append({k: kw[k].detach().clone() for k in FIELDS if k in kw and kw[k] is not None})
        return self.model(**kw)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    consts = synthetic.make_body_model_x(seed=0)
    model = TorchSMPLX(consts)
    gmm = synthetic.make_gmm(seed=0)
    scratch = tempfile.mkdtemp(prefix="k2b_goldenx_")
    os.makedirs(os.path.join(scratch, "data", "models"))
    with open(os.path.join(scratch, "data", "models", "gmm_08.pkl"), "wb") as f:
        pickle.dump({"means": gmm.means, "covars": gmm.covars, "weights": gmm.weights}, f)
    os.chdir(scratch)
    WorldSpaceFitter, _, _, _ = import_reference()
    from keypoints2body.models.smpl_data import SMPLXData  # type: ignore

    T = 4
    poses = synthetic.make_poses_x(T, seed=0)
    tt = lambda a: torch.tensor(np.asarray(a))
    truth = {k: tt(getattr(poses, k)) for k in FIELDS}
    with torch.no_grad():
        gt = model(**truth)
    # targets: the 22 body joints + jaw, eyes + both hands = all 55 kinematic joints; and the AMASS-22 subset
    j55 = gt.joints[:, :55].clone() + tt(synthetic.target_noise(T, 55, seed=3, scale=0.003))
    idx55 = torch.arange(55, dtype=torch.long)

    only = sys.argv[1] if len(sys.argv) > 1 else None      # python oracle/gen_golden_smplx.py [case]: one case only

    def run(name, init, j3d, conf, seq_ind, num_iters, category, target_model_indices, freeze_betas=False):
        if only is not None and name != only:
            return
        B = j3d.shape[0]
        rec = RecorderX(model)
        fitter = WorldSpaceFitter(rec, step_size=1e-2, num_iters_first=num_iters if seq_ind == 0 else 7,
                                  num_iters_followup=num_iters if seq_ind > 0 else 7, use_lbfgs=False,
                                  joints_category=category, device=torch.device("cpu"))
        fitter.pose_prior = PaddedPrior(fitter.pose_prior)
        trace_iters = [t for t in TRACE_ITERS if t <= num_iters]
        out = {k: [] for k in FIELDS + ("joints", "verts_sampled", "loss")}
        trace = {k: [[] for _ in trace_iters] for k in FIELDS}
        iter_losses = []
        for i in range(B):
            sl = slice(i, i + 1)
            rec.snaps.clear()
            losses = []
            orig = torch.Tensor.backward

            def spy(self, *a, **k):
                losses.append(float(self.detach()))
                return orig(self, *a, **k)

            torch.Tensor.backward = spy
            try:
                res = fitter.fit_frame(SMPLXData(**{k: init[k][sl] for k in FIELDS}), j3d[sl], conf_3d=conf, seq_ind=seq_ind,
                                       target_model_indices=target_model_indices, joint_loss_weight=600.0,
                                       pose_preserve_weight=5.0, freeze_betas=freeze_betas)
            finally:
                torch.Tensor.backward = orig
            assert len(rec.snaps) == num_iters + 1 and len(losses) == num_iters
            iter_losses.append(losses)
            for ti, t in enumerate(trace_iters):
                for k in FIELDS:
                    trace[k][ti].append(rec.snaps[t][k])
            for k in FIELDS:
                out[k].append(getattr(res.params, k))
            out["joints"].append(res.joints)
            out["verts_sampled"].append(res.vertices[:, sample_vertex_ids(res.vertices.shape[1])])
            out["loss"].append(res.loss.reshape(1))
        cat = lambda xs: torch.cat(xs, dim=0).detach().numpy()
        payload = dict(case=name, category=category, seq_ind=seq_ind, num_iters=num_iters, freeze_betas=int(freeze_betas),
                       model_fingerprint=np.uint64(consts.fingerprint()), j3d=j3d.numpy(),
                       conf=(conf.numpy() if conf is not None else np.zeros(0, np.float32)), has_conf=int(conf is not None),
                       target_model_indices=(target_model_indices.numpy() if target_model_indices is not None else np.zeros(0, np.int64)),
                       trace_iters=np.array(trace_iters), iter_losses=np.array(iter_losses),
                       sampled_vertex_ids=sample_vertex_ids(consts.num_vertices),
                       out_joints=cat(out["joints"]), out_verts_sampled=cat(out["verts_sampled"]), out_loss=cat(out["loss"]))
        for k in FIELDS:
            payload["init_" + k] = init[k][:B].numpy()
            payload["out_" + k] = cat(out[k])
            payload["trace_" + k] = np.stack([cat(trace[k][ti]) for ti in range(len(trace_iters))])
        np.savez_compressed(GOLDEN / f"smplx_fit_{name}.npz", **payload)
        print(f"[golden] smplx {name}: B={B} iters={num_iters} losses={payload['out_loss']}")

    zeros = lambda c: torch.zeros(T, c)
    with torch.no_grad():
        j0 = model(global_orient=zeros(3), body_pose=zeros(63)).joints
    zero_init = dict(global_orient=zeros(3), body_pose=zeros(63), transl=(j55[:, 0] - j0[:, 0]).detach() + 0.01,
                     left_hand_pose=zeros(45), right_hand_pose=zeros(45), expression=zeros(10), jaw_pose=zeros(3),
                     leye_pose=zeros(3), reye_pose=zeros(3), betas=zeros(10))
    # 1. all 55 kinematic joints observed (hands and face fitted), zero init, 100 iterations
    conf55 = torch.ones(55); conf55[[7, 8, 10, 11]] = 1.5; conf55[[30, 45]] = 0.7
    run("all55_zero_init", zero_init, j55[:3], conf55, 0, 100, "GENERIC", idx55)
    # 2. body joints only (AMASS-22 category): hands / face parameters get no gradient and must stay put
    run("amass22_zero_init", zero_init, j55[:2, :22].clone(), None, 0, 50, "AMASS", None)
    # 3. follow-up frame (preserve term on the 63 body dimensions), warm start, frozen betas (expression stays free)
    warm = {k: truth[k] * 0.8 for k in FIELDS}
    warm["transl"] = truth["transl"] + 0.02
    run("all55_followup_frozen", warm, j55[:2], conf55, 3, 30, "GENERIC", idx55, freeze_betas=True)
    # 4. vertex-selected joints among the targets (smplx "extra" joints = single mesh vertices, model indices >= 55): the loss
    #    differentiates through blend shapes and skinning of those vertices (world_space.py:198-201)
    idxv = torch.tensor(list(range(55)) + [55, 61, 75, 100, 126], dtype=torch.long)
    jv = gt.joints[:, idxv].clone() + tt(synthetic.target_noise(T, 60, seed=5, scale=0.003))
    confv = torch.ones(60); confv[[7, 8, 10, 11]] = 1.5; confv[[56, 58]] = 0.8
    run("vertex_joints_zero_init", zero_init, jv[:2], confv, 0, 30, "GENERIC", idxv)
    print("smplx goldens written to", GOLDEN)


if __name__ == "__main__":
    main()
