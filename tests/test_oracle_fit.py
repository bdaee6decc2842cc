"""Pins the oracle (oracle/fit_torch.py) to the golden vectors produced by the REAL
reference fitter (oracle/gen_golden.py, run in the build container)."""
import numpy as np
import pytest
import torch

from oracle.fit_torch import fit_world_adam
from tests import helpers as H

TOL = 2e-6   # same arithmetic, same library: differences are batch-shape / thread effects only


def test_prior_known_answers():
    d = np.load(H.GOLDEN / "prior_probe.npz")
    got = H.oracle_prior()(torch.tensor(d["pose"]), None).numpy()
    np.testing.assert_allclose(got, d["value"], rtol=2e-6)


def test_prior_buffers_rebuilt_from_mixture_match_reference():
    from oracle.fit_torch import GMMPrior
    g = H.gmm_fixture()
    p = GMMPrior(g["means"], g["covars"].astype(np.float64), g["weights"])
    assert np.abs(p.precisions.numpy() - g["ref_precisions"]).max() <= 1e-4 * np.abs(g["ref_precisions"]).max()
    np.testing.assert_allclose(p.nll_weights.numpy(), g["ref_nll_weights"], rtol=1e-5)   # covars stored as f32


@pytest.mark.parametrize("case", H.WORLD_CASES + ("generic_vertex_joints",))
def test_oracle_reproduces_reference_fit(case):
    d = H.load_case(case)
    t = lambda k: torch.tensor(d[k])
    idx = H.case_indices(d)
    conf = t("conf") if int(d["has_conf"]) else None
    out = fit_world_adam(H.oracle_model(), H.oracle_prior(), t("init_global_orient"), t("init_body_pose"),
                         t("init_betas"), t("init_transl"), t("j3d"), conf, num_iters=int(d["num_iters"]),
                         seq_ind=int(d["seq_ind"]), model_idx=idx, freeze_betas=bool(int(d["freeze_betas"])),
                         trace_iters=tuple(int(i) for i in d["trace_iters"]), record_all_losses=True)
    for ti in range(len(d["trace_iters"])):
        assert np.abs(out.trace.global_orient[ti].numpy() - d["trace_global_orient"][ti]).max() < TOL
        assert np.abs(out.trace.body_pose[ti].numpy() - d["trace_body_pose"][ti]).max() < TOL
        assert np.abs(out.trace.betas[ti].numpy() - d["trace_betas"][ti]).max() < TOL
        assert np.abs(out.trace.transl[ti].numpy() - d["trace_transl"][ti]).max() < TOL
    for k, o in (("global_orient", out.global_orient), ("body_pose", out.body_pose), ("betas", out.betas),
                 ("transl", out.transl), ("joints", out.joints)):
        assert np.abs(o.numpy() - d["out_" + k]).max() < TOL
    vid = torch.as_tensor(d["sampled_vertex_ids"])
    assert np.abs(out.vertices[:, vid].numpy() - d["out_verts_sampled"]).max() < TOL
    # per-iteration losses the reference back-propagated (per call: sum over the call's batch)
    mine = torch.stack(out.trace.loss, dim=0).double().numpy()          # (iters, B)
    ref = d["iter_losses"]                                               # (calls, iters)
    if int(d["per_frame_calls"]):
        np.testing.assert_allclose(mine.T, ref, rtol=1e-5)
    else:
        np.testing.assert_allclose(mine.sum(axis=1)[None], ref, rtol=1e-5)
    if int(d["freeze_betas"]):
        assert np.array_equal(out.betas.numpy(), d["init_betas"])


CAMERA_CASES = ("full", "followup_frozen")


@pytest.mark.parametrize("case", CAMERA_CASES)
def test_oracle_reproduces_reference_camera_fit(case):
    """The camera-space restatement (two Adam stages) vs the reference's CameraSpaceFitter."""
    from oracle.fit_torch import fit_camera_adam_one
    d = dict(np.load(H.GOLDEN / f"camera_fit_{case}.npz"))
    conf = torch.tensor(d["conf"]) if int(d["has_conf"]) else None
    for i in range(d["j3d"].shape[0]):
        t = lambda k: torch.tensor(d[k][i:i + 1])
        o = fit_camera_adam_one(H.oracle_model(), H.oracle_prior(), t("init_global_orient"), t("init_body_pose"),
                                t("init_betas"), t("j3d"), conf, num_iters=int(d["num_iters"]),
                                seq_ind=int(d["seq_ind"]), freeze_betas=bool(int(d["freeze_betas"])),
                                init_cam_t=t("init_cam_t") if int(d["has_init_cam_t"]) else None)
        for key, val in (("global_orient", o.global_orient), ("body_pose", o.body_pose), ("betas", o.betas),
                         ("transl", o.transl), ("joints", o.joints)):
            assert np.abs(val.numpy() - d["out_" + key][i:i + 1]).max() < 5e-6, (case, i, key)
        np.testing.assert_allclose(float(o.loss), float(d["out_loss"][i]), rtol=1e-5)


@pytest.mark.parametrize("case", ["first", "followup", "frozen"])
def test_oracle_reproduces_reference_lbfgs_fit(case):
    """LBFGS branch of the world fitter (the reference's default) vs the reference's own run."""
    from oracle.fit_torch import fit_world_lbfgs_one
    d = dict(np.load(H.GOLDEN / f"lbfgs_world_{case}.npz"))
    for i in range(d["j3d"].shape[0]):
        t = lambda k: torch.tensor(d[k][i:i + 1])
        o = fit_world_lbfgs_one(H.oracle_model(), H.oracle_prior(), t("init_global_orient"), t("init_body_pose"),
                                t("init_betas"), t("init_transl"), t("j3d"), torch.tensor(d["conf"]),
                                max_iter=int(d["max_iter"]), seq_ind=int(d["seq_ind"]),
                                freeze_betas=bool(int(d["freeze_betas"])))
        for key, val in (("global_orient", o.global_orient), ("body_pose", o.body_pose), ("betas", o.betas),
                         ("transl", o.transl)):
            assert np.abs(val.numpy() - d["out_" + key][i:i + 1]).max() < 2e-5, (case, i, key)
        np.testing.assert_allclose(float(o.loss), float(d["out_loss"][i]), rtol=1e-5)


def test_oracle_reproduces_reference_shape_pass():
    from oracle.fit_torch import shape_pass_lbfgs
    d = dict(np.load(H.GOLDEN / "shape_pass.npz"))
    T = d["j3d"].shape[0]
    betas = shape_pass_lbfgs(H.oracle_model(), torch.tensor(d["init_betas"]), torch.tensor(d["mean_pose"]).repeat(T, 1),
                             torch.tensor(d["j3d"]), torch.tensor(d["conf"]), list(range(int(d["num_shape_frames"]))),
                             num_iters=int(d["num_shape_iters"]))
    assert np.abs(betas.numpy() - d["out_betas"]).max() < 2e-5
    # (the joint term has weight 1 here against 25 |beta|^2 per frame, so the reference's pre-pass
    #  returns betas ~ 0 even for targets generated with a distinct shape: pinned as it behaves)


SMPLX_CASES = ("all55_zero_init", "amass22_zero_init", "all55_followup_frozen", "vertex_joints_zero_init")


@pytest.mark.parametrize("case", SMPLX_CASES)
def test_oracle_reproduces_reference_smplx_fit(case):
    """SMPL-X goldens (oracle/gen_golden_smplx.py: the real reference fitter with SMPLXData, 55-joint tree, its prior
    evaluated at [body_pose | 0 x 6]) against the oracle's SMPL-X restatement, at every recorded iteration."""
    from oracle.fit_torch import SMPLX_FIELDS, fit_world_adam_smplx
    d = H.load_smplx_case(case)
    t = lambda k: torch.tensor(d[k])
    idx = [int(i) for i in d["target_model_indices"]] if d["target_model_indices"].size else list(range(22))
    params, loss, joints, verts, trace = fit_world_adam_smplx(
        H.oracle_model_x(), H.oracle_prior(), {k: t("init_" + k) for k in SMPLX_FIELDS}, t("j3d"),
        t("conf") if int(d["has_conf"]) else None, num_iters=int(d["num_iters"]), seq_ind=int(d["seq_ind"]), model_idx=idx,
        freeze_betas=bool(int(d["freeze_betas"])), trace_iters=tuple(int(i) for i in d["trace_iters"]))
    for ti, it in enumerate(d["trace_iters"]):
        for k in SMPLX_FIELDS:
            assert np.abs(trace[int(it)][k].numpy() - d["trace_" + k][ti]).max() < TOL, (case, int(it), k)
    for k in SMPLX_FIELDS:
        assert np.abs(params[k].numpy() - d["out_" + k]).max() < TOL, (case, k)
    assert np.abs(joints.numpy() - d["out_joints"]).max() < TOL
    assert np.abs(verts[:, torch.as_tensor(d["sampled_vertex_ids"])].numpy() - d["out_verts_sampled"]).max() < TOL
    np.testing.assert_allclose(loss.numpy(), d["iter_losses"][:, -1], rtol=1e-5)
    if case == "amass22_zero_init":          # no target on hands / face: their poses never move (the expression does: it is a
        for k in ("left_hand_pose", "right_hand_pose", "jaw_pose", "leye_pose", "reye_pose"):   # shape coefficient of every joint)
            assert np.array_equal(params[k].numpy(), d["init_" + k])
    if int(d["freeze_betas"]):
        assert np.array_equal(params["betas"].numpy(), d["init_betas"])
        assert np.abs(params["expression"].numpy() - d["init_expression"]).max() > 1e-4    # the expression stays free


@pytest.mark.parametrize("case", ("all52_zero_init", "all52_followup_frozen"))
def test_oracle_reproduces_reference_smplh_fit(case):
    """SMPL-H goldens (oracle/gen_golden_smplh.py: the real reference fitter with SMPLHData, 52-joint tree, both hands in the
    optimiser) against the oracle's restatement, at every recorded iteration."""
    from oracle.fit_torch import SMPLH_FIELDS, fit_world_adam_smplx
    d = H.load_smplh_case(case)
    t = lambda k: torch.tensor(d[k])
    idx = [int(i) for i in d["target_model_indices"]]
    params, loss, joints, verts, trace = fit_world_adam_smplx(
        H.oracle_model_h(), H.oracle_prior(), {k: t("init_" + k) for k in SMPLH_FIELDS}, t("j3d"), t("conf"),
        num_iters=int(d["num_iters"]), seq_ind=int(d["seq_ind"]), model_idx=idx, freeze_betas=bool(int(d["freeze_betas"])),
        trace_iters=tuple(int(i) for i in d["trace_iters"]), fields=SMPLH_FIELDS)
    for ti, it in enumerate(d["trace_iters"]):
        for k in SMPLH_FIELDS:
            assert np.abs(trace[int(it)][k].numpy() - d["trace_" + k][ti]).max() < TOL, (case, int(it), k)
    for k in SMPLH_FIELDS:
        assert np.abs(params[k].numpy() - d["out_" + k]).max() < TOL, (case, k)
    assert np.abs(joints.numpy() - d["out_joints"]).max() < TOL
    assert np.abs(verts[:, torch.as_tensor(d["sampled_vertex_ids"])].numpy() - d["out_verts_sampled"]).max() < TOL
    np.testing.assert_allclose(loss.numpy(), d["iter_losses"][:, -1], rtol=1e-5)
    if int(d["freeze_betas"]):
        assert np.array_equal(params["betas"].numpy(), d["init_betas"])


@pytest.mark.parametrize("name", H.CHAIN_CASES)
def test_oracle_chain_matches_reference_sequence_loop_on_real_motion(name):
    """The oracle restatement walked through the reference's frame loop (api/sequence.py:214-281: per-frame fix_foot
    confidences, ``seq_ind = idx``, every frame starting from - and preserving - its predecessor's RESULT) against what the
    real reference produced on the demo motions (fixtures of oracle/gen_golden_chain.py).

    Two gates.  TEACHER-FORCED, 1e-4 on every frame: frame t fitted from the REFERENCE's result of frame t-1 - pins the
    per-frame semantics of the loop without error accumulation.  FREE-RUNNING: the chain amplifies rounding (the reference
    walked again from a start perturbed by 2e-6 relative ends up to 2.6e-2 away from itself after 14 frames:
    ``out_param_dev_perturbed`` in the fixture), so frame t may deviate by max(1e-4, 4 x the reference's own deviation)."""
    from oracle.fit_torch import fit_world_adam
    torch.set_num_threads(8)
    d = H.load_chain_case(name)
    model, prior = H.oracle_model(), H.oracle_prior()
    t = lambda k: torch.tensor(d[k])
    keys = ("global_orient", "body_pose", "betas", "transl")
    j3d, conf = t("j3d"), t("conf")
    iters = lambda i: int(d["num_iters_first"] if i == 0 else d["num_iters_followup"])
    cur = tuple(t("init_" + k) for k in keys)
    for i in range(j3d.shape[0]):
        # free-running
        o = fit_world_adam(model, prior, *cur, j3d[i:i + 1], conf[i], seq_ind=i, num_iters=iters(i))
        got = (o.global_orient, o.body_pose, o.betas, o.transl)
        dev = max(float((v - t("out_" + k)[i:i + 1]).abs().max()) for k, v in zip(keys, got))
        assert dev < max(1e-4, 4.0 * float(d["out_param_dev_perturbed"][i])), (i, dev)
        cur = got
        # teacher-forced
        start = cur if i == 0 else tuple(t("out_" + k)[i - 1:i] for k in keys)
        if i > 0:
            o = fit_world_adam(model, prior, *start, j3d[i:i + 1], conf[i], seq_ind=i, num_iters=iters(i))
        for k, v in zip(keys, (o.global_orient, o.body_pose, o.betas, o.transl)):
            assert float((v - t("out_" + k)[i:i + 1]).abs().max()) < 1e-4, (i, k)
        assert abs(float(o.loss) - float(d["out_loss"][i])) <= 1e-4 * abs(float(d["out_loss"][i])), i
        assert float((o.joints - t("out_joints")[i:i + 1]).abs().max()) < 1e-4, i
