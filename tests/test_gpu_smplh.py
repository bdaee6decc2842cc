"""SMPL-H (52 joints: body + two 15-joint hands) on the GPU, through the C ABI and the reference-shaped fitter.

Pinned by ``tests/golden/smplh_fit_*.npz``: the REAL reference fitter driven with ``SMPLHData`` and the oracle's SMPL-H model
(``oracle/gen_golden_smplh.py``; prior evaluated at [body_pose | 0 x 6] as for SMPL-X).  PARITY UNPINNED at the
smplx boundary.  Tolerances: fitted parameters 1e-4 abs at every recorded iteration, LBS 5e-6 m.
"""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
PARAM_TOL = 1e-4
POSE_FIELDS = (("body_pose", 63), ("left_hand_pose", 45), ("right_hand_pose", 45))
FIELDS = ("global_orient", "body_pose", "transl", "left_hand_pose", "right_hand_pose", "betas")


def pack(d, prefix):
    return d[prefix + "global_orient"], np.concatenate([d[prefix + k] for k, _ in POSE_FIELDS], axis=1), d[prefix + "betas"], d[prefix + "transl"]


def native_fit_h(d, num_iters):
    from keypoints2body_amd import native
    cfg = native.default_fit_config()
    cfg.num_iters = int(num_iters)
    cfg.pose_preserve_weight = 5.0 if int(d["seq_ind"]) > 0 else 0.0
    cfg.freeze_betas = int(d["freeze_betas"])
    cfg.prior_pose_dims, cfg.num_betas_prior = 63, 10
    idx = [int(i) for i in d["target_model_indices"]]
    go, pose, shape, tr = map(H.cuda, pack(d, "init_"))
    return native.fit_world(H.native_model_h(), H.native_prior(), cfg, idx, H.cuda(d["j3d"]), H.cuda(d["conf"]), go, pose, shape, tr)


def test_smplh_lbs_matches_oracle_forward():
    from keypoints2body_amd import synthetic
    B = 41
    p = synthetic.make_poses_h(B, seed=4)
    t = lambda a: torch.tensor(np.asarray(a))
    with torch.no_grad():
        ref = H.oracle_model_h()(**{k: t(getattr(p, k)) for k in FIELDS})
    pose = np.concatenate([getattr(p, k) for k, _ in POSE_FIELDS], axis=1)
    j, v = H.native_model_h().lbs(H.cuda(p.global_orient), H.cuda(pose), H.cuda(p.betas), H.cuda(p.transl))
    assert tuple(j.shape) == (B, 52 + 21, 3) and tuple(v.shape) == (B, 6890, 3)
    assert (v.cpu() - ref.vertices).abs().max() < 5e-6
    assert (j.cpu() - ref.joints).abs().max() < 5e-6


@pytest.mark.parametrize("case", ("all52_zero_init", "all52_followup_frozen"))
def test_smplh_fit_matches_reference_golden(case):
    d = H.load_smplh_case(case)
    worst = 0.0
    for ti, it in enumerate(d["trace_iters"]):
        out = native_fit_h(d, it)
        pose = np.concatenate([d["trace_" + k][ti] for k, _ in POSE_FIELDS], axis=1)
        for key, want in (("global_orient", d["trace_global_orient"][ti]), ("body_pose", pose), ("betas", d["trace_betas"][ti]),
                          ("transl", d["trace_transl"][ti])):
            err = np.abs(out[key].cpu().numpy() - want).max()
            worst = max(worst, err)
            assert err < PARAM_TOL, f"{case} iteration {int(it)}: {key} differs by {err}"
        np.testing.assert_allclose(out["loss"].cpu().numpy(), d["iter_losses"][:, int(it) - 1], rtol=2e-4)
    out = native_fit_h(d, d["num_iters"])
    go, pose, shape, tr = pack(d, "out_")
    for key, want in (("global_orient", go), ("body_pose", pose), ("betas", shape), ("transl", tr)):
        assert np.abs(out[key].cpu().numpy() - want).max() < PARAM_TOL, (case, key)
    j, v = H.native_model_h().lbs(out["global_orient"], out["body_pose"], out["betas"], out["transl"])
    assert np.abs(j.cpu().numpy() - d["out_joints"]).max() < PARAM_TOL
    assert np.abs(v[:, torch.as_tensor(d["sampled_vertex_ids"]).cuda()].cpu().numpy() - d["out_verts_sampled"]).max() < PARAM_TOL
    if int(d["freeze_betas"]):
        assert torch.equal(out["betas"].cpu(), torch.tensor(d["init_betas"]))
    print(f"smplh {case}: worst parameter deviation over the trace = {worst:.2e}")


def test_smplh_fitter_and_sequence_api_return_smplh_data():
    """WorldSpaceFitter.fit_frame with SMPLHData in -> SMPLHData out (both hands fitted); optimize_params_sequence with
    body_model="smplh" runs the warm-start chain as one launch and returns the same bits as frame-by-frame calls."""
    import keypoints2body_amd as k2b
    from keypoints2body_amd.core.fitters.world_space import WorldSpaceFitter
    from keypoints2body_amd.models.body_model import BodyModel
    from keypoints2body_amd.models.smpl_data import SMPLHData
    from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers
    d = H.load_smplh_case("all52_zero_init")
    g = H.gmm_fixture()
    c = H.body_consts_h()
    model = BodyModel(c.v_template, c.shapedirs, c.posedirs, c.J_regressor, c.lbs_weights, c.parents, c.extra_vertex_ids)
    assert model.model_type == "smplh" and model.packed and model.num_betas == 10
    prior = MaxMixturePrior(MixtureBuffers(g["ref_means"], g["ref_precisions"], g["ref_nll_weights"].reshape(-1)))
    fitter = WorldSpaceFitter(model, step_size=1e-2, num_iters_first=int(d["num_iters"]), num_iters_followup=8, use_lbfgs=False,
                              joints_category="GENERIC", pose_prior=prior)
    init = SMPLHData(**{k: torch.tensor(d["init_" + k][:1]) for k in FIELDS})
    idx = torch.tensor(d["target_model_indices"])
    res = fitter.fit_frame(init, torch.tensor(d["j3d"][:1]), conf_3d=torch.tensor(d["conf"]), seq_ind=0, target_model_indices=idx)
    assert isinstance(res.params, SMPLHData)
    for k in FIELDS:
        assert np.abs(getattr(res.params, k).cpu().numpy() - d["out_" + k][:1]).max() < PARAM_TOL, k
    assert float(res.params.left_hand_pose.abs().max()) > 0.0
    # sequence API on the AMASS-22 subset of the same frames (52-joint model, hands untargeted)
    seq = d["j3d"][:3, :22]
    cfg = {"frame": {"use_lbfgs": False, "num_iters_first": 10, "num_iters_followup": 5}, "use_shape_optimization": False,
           "fix_foot": False}
    out = k2b.optimize_params_sequence(seq, init_params=init, body_model="smplh", joint_layout="AMASS", model=model, config=cfg,
                                       pose_prior=prior, mean_params=(torch.zeros(1, 66), torch.zeros(1, 10)))
    f2 = WorldSpaceFitter(model, step_size=1e-2, num_iters_first=10, num_iters_followup=5, use_lbfgs=False, joints_category="AMASS",
                          pose_prior=prior)
    prev = init
    for i in range(3):
        want = f2.fit_frame(prev, torch.tensor(seq[i:i + 1]), conf_3d=torch.ones(22), seq_ind=i)
        assert isinstance(out[i].params, SMPLHData)
        for k in FIELDS:
            assert torch.equal(getattr(out[i].params, k), getattr(want.params, k)), (i, k)
        prev = want.params


@pytest.mark.parametrize("name", ["smplh", "smplx"])
def test_packed_models_through_the_public_api_with_default_configurations(name):
    """optimize_params_frame / optimize_params_sequence with their DEFAULT configurations (L-BFGS, shape pre-pass, warm start) and
    a 52- / 55-joint model: the reference's data class for the model comes back, finite, with the full mesh."""
    import keypoints2body_amd as k2b
    from keypoints2body_amd.models.body_model import BodyModel
    from keypoints2body_amd.models.smpl_data import SMPLHData, SMPLXData
    from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers
    g = H.gmm_fixture()
    prior = MaxMixturePrior(MixtureBuffers(g["ref_means"], g["ref_precisions"], g["ref_nll_weights"].reshape(-1)))
    model, cls, V = (BodyModel.synthetic_h(0), SMPLHData, 6890) if name == "smplh" else (BodyModel.synthetic_x(0), SMPLXData, 10475)
    j3d = H.load_smplx_case("amass22_zero_init")["j3d"]
    mean = (torch.zeros(1, 66), torch.zeros(1, 10))
    for cfg in ({}, {"use_lbfgs": False}):
        r = k2b.optimize_params_frame(j3d[0], body_model=name, joint_layout="AMASS", model=model, config=cfg, pose_prior=prior,
                                      mean_params=mean)
        assert isinstance(r.params, cls) and tuple(r.vertices.shape) == (1, V, 3) and torch.isfinite(r.loss)
    rs = k2b.optimize_params_sequence(np.tile(j3d, (2, 1, 1))[:3], body_model=name, joint_layout="AMASS", model=model, pose_prior=prior,
                                      mean_params=mean)
    assert len(rs) == 3 and all(isinstance(x.params, cls) and torch.isfinite(x.loss) for x in rs)
