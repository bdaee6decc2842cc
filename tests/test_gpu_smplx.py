"""SMPL-X (BASELINE config 4: 55 joints, hands + face) on the GPU, through the C ABI.

Pinned by ``tests/golden/smplx_fit_*.npz``: the REAL reference fitter driven with ``SMPLXData`` and the oracle's SMPL-X
model (``oracle/gen_golden_smplx.py``; its 69-D prior evaluated at [body_pose | 0 x 6], the one definition this engine
adds - SURVEY.md N3).  PARITY UNPINNED at the smplx boundary (the forward restates the published formulation), as for SMPL.
Tolerances: fitted parameters 1e-4 abs (the north-star bar) at every recorded iteration, LBS 5e-6 m.
"""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
PARAM_TOL = 1e-4
CASES = ("all55_zero_init", "amass22_zero_init", "all55_followup_frozen", "vertex_joints_zero_init")
POSE_FIELDS = (("body_pose", 63), ("jaw_pose", 3), ("leye_pose", 3), ("reye_pose", 3), ("left_hand_pose", 45), ("right_hand_pose", 45))


def pack(d, prefix, rows=slice(None)):
    pose = np.concatenate([d[prefix + k][rows] for k, _ in POSE_FIELDS], axis=1)
    shape = np.concatenate([d[prefix + "betas"][rows], d[prefix + "expression"][rows]], axis=1)
    return d[prefix + "global_orient"][rows], pose, shape, d[prefix + "transl"][rows]


def native_fit_x(d, num_iters, shape=0):
    from keypoints2body_amd import native
    cfg = native.default_fit_config()
    cfg.debug_launch_shape = shape                  # tree kernel: 0 = by batch size, 1 = plain, 2 = component waves
    cfg.num_iters = int(num_iters)
    cfg.pose_preserve_weight = 5.0 if int(d["seq_ind"]) > 0 else 0.0
    cfg.freeze_betas = int(d["freeze_betas"])
    cfg.prior_pose_dims, cfg.num_betas_prior = 63, 10
    idx = [int(i) for i in d["target_model_indices"]] if d["target_model_indices"].size else list(range(22))
    go, pose, shape, tr = map(H.cuda, pack(d, "init_"))
    conf = H.cuda(d["conf"]) if int(d["has_conf"]) else None
    return native.fit_world(H.native_model_x(), H.native_prior(), cfg, idx, H.cuda(d["j3d"]), conf, go, pose, shape, tr)


def test_smplx_lbs_matches_oracle_forward():
    from keypoints2body_amd import synthetic
    B = 37                                              # ragged: one full and one partial 32-frame tile
    p = synthetic.make_poses_x(B, seed=4)
    t = lambda a: torch.tensor(np.asarray(a))
    with torch.no_grad():
        ref = H.oracle_model_x()(**{k: t(getattr(p, k)) for k in ("global_orient", "body_pose", "jaw_pose", "leye_pose", "reye_pose",
                                                                   "left_hand_pose", "right_hand_pose", "betas", "expression", "transl")})
    pose = np.concatenate([getattr(p, k) for k, _ in POSE_FIELDS], axis=1)
    shape = np.concatenate([p.betas, p.expression], axis=1)
    j, v = H.native_model_x().lbs(H.cuda(p.global_orient), H.cuda(pose), H.cuda(shape), H.cuda(p.transl))
    assert tuple(j.shape) == (B, 127, 3) and tuple(v.shape) == (B, 10475, 3)
    assert (v.cpu() - ref.vertices).abs().max() < 5e-6
    assert (j.cpu() - ref.joints).abs().max() < 5e-6
    j2, _ = H.native_model_x().lbs(H.cuda(p.global_orient), H.cuda(pose), H.cuda(shape), H.cuda(p.transl), want_vertices=False)
    assert (j2.cpu() - ref.joints).abs().max() < 5e-6


@pytest.mark.parametrize("B", [129, 300])
def test_smplx_lbs_ragged_multi_tile_batches(B):
    """Several 128-frame groups with a partial last one: the persistent workgroups of the SMPL-X skinning kernel walk more than
    one tile (operands of the next tile fetched under the current one) and the last frame group takes the predicated stores."""
    from keypoints2body_amd import synthetic
    p = synthetic.make_poses_x(B, seed=11)
    t = lambda a: torch.tensor(np.asarray(a))
    with torch.no_grad():
        ref = H.oracle_model_x()(**{k: t(getattr(p, k)) for k in ("global_orient", "body_pose", "jaw_pose", "leye_pose", "reye_pose",
                                                                   "left_hand_pose", "right_hand_pose", "betas", "expression", "transl")})
    pose = np.concatenate([getattr(p, k) for k, _ in POSE_FIELDS], axis=1)
    shape = np.concatenate([p.betas, p.expression], axis=1)
    j, v = H.native_model_x().lbs(H.cuda(p.global_orient), H.cuda(pose), H.cuda(shape), H.cuda(p.transl))
    assert (v.cpu() - ref.vertices).abs().max() < 5e-6
    assert (j.cpu() - ref.joints).abs().max() < 5e-6


@pytest.mark.parametrize("case", CASES)
def test_smplx_fit_matches_reference_golden(case):
    d = H.load_smplx_case(case)
    worst = 0.0
    for ti, it in enumerate(d["trace_iters"]):
        out = native_fit_x(d, it)
        pose = np.concatenate([d["trace_" + k][ti] for k, _ in POSE_FIELDS], axis=1)
        shape = np.concatenate([d["trace_betas"][ti], d["trace_expression"][ti]], axis=1)
        for key, want in (("global_orient", d["trace_global_orient"][ti]), ("body_pose", pose), ("betas", shape), ("transl", d["trace_transl"][ti])):
            err = np.abs(out[key].cpu().numpy() - want).max()
            worst = max(worst, err)
            assert err < PARAM_TOL, f"{case} iteration {int(it)}: {key} differs by {err}"
        np.testing.assert_allclose(out["loss"].cpu().numpy(), d["iter_losses"][:, int(it) - 1], rtol=2e-4, err_msg=f"{case} it {int(it)}")
    out = native_fit_x(d, d["num_iters"])
    go, pose, shape, tr = pack(d, "out_")
    for key, want in (("global_orient", go), ("body_pose", pose), ("betas", shape), ("transl", tr)):
        assert np.abs(out[key].cpu().numpy() - want).max() < PARAM_TOL, (case, key)
    j, v = H.native_model_x().lbs(out["global_orient"], out["body_pose"], out["betas"], out["transl"])
    assert np.abs(j.cpu().numpy() - d["out_joints"]).max() < PARAM_TOL
    assert np.abs(v[:, torch.as_tensor(d["sampled_vertex_ids"]).cuda()].cpu().numpy() - d["out_verts_sampled"]).max() < PARAM_TOL
    if int(d["freeze_betas"]):
        assert torch.equal(out["betas"][:, :10].cpu(), torch.tensor(d["init_betas"]))
    print(f"smplx {case}: worst parameter deviation over the trace = {worst:.2e}")


@pytest.mark.parametrize("case", CASES)
def test_smplx_both_shapes_of_the_tree_kernel_agree_and_match_the_golden(case):
    """The tree kernel runs the mixture either inside every frame wave (plain shape: more than four frames per CU) or on four
    dedicated component waves beside the frame waves (at most four frames per CU - every small case above).  Same arithmetic
    in the same order: the two shapes must agree to the last bit, and each reproduces the reference's end state."""
    d = H.load_smplx_case(case)
    plain, comp = native_fit_x(d, d["num_iters"], shape=1), native_fit_x(d, d["num_iters"], shape=2)
    go, pose, shape, tr = pack(d, "out_")
    for key, want in (("global_orient", go), ("body_pose", pose), ("betas", shape), ("transl", tr)):
        assert torch.equal(plain[key], comp[key]), (case, key)
        assert np.abs(plain[key].cpu().numpy() - want).max() < PARAM_TOL, (case, key)
    assert torch.equal(plain["loss"], comp["loss"])


def test_smplx_fitter_api_returns_smplx_data():
    """The reference-shaped entry: WorldSpaceFitter.fit_frame with SMPLXData in -> SMPLXData out, all ten fields fitted."""
    from keypoints2body_amd.core.fitters.world_space import WorldSpaceFitter
    from keypoints2body_amd.models.body_model import BodyModel
    from keypoints2body_amd.models.smpl_data import SMPLXData
    from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers
    d = H.load_smplx_case("all55_zero_init")
    g = H.gmm_fixture()
    c = H.body_consts_x()
    model = BodyModel(c.v_template, c.shapedirs, c.posedirs, c.J_regressor, c.lbs_weights, c.parents, c.extra_vertex_ids)
    assert model.model_type == "smplx" and model.num_betas == 10 and model.num_expression_coeffs == 10
    prior = MaxMixturePrior(MixtureBuffers(g["ref_means"], g["ref_precisions"], g["ref_nll_weights"].reshape(-1)))
    fitter = WorldSpaceFitter(model, step_size=1e-2, num_iters_first=int(d["num_iters"]), use_lbfgs=False,
                              joints_category="GENERIC", pose_prior=prior)
    fields = ("global_orient", "body_pose", "transl", "left_hand_pose", "right_hand_pose", "expression", "jaw_pose", "leye_pose",
              "reye_pose", "betas")
    init = SMPLXData(**{k: torch.tensor(d["init_" + k][:1]) for k in fields})
    res = fitter.fit_frame(init, torch.tensor(d["j3d"][:1]), conf_3d=torch.tensor(d["conf"]), seq_ind=0,
                           target_model_indices=torch.tensor(d["target_model_indices"]))
    assert isinstance(res.params, SMPLXData)
    for k in fields:
        assert np.abs(getattr(res.params, k).cpu().numpy() - d["out_" + k][:1]).max() < PARAM_TOL, k
    assert tuple(res.joints.shape) == (1, 127, 3) and tuple(res.vertices.shape) == (1, 10475, 3)
    assert abs(float(res.loss) - float(d["out_loss"][0])) < 2e-4 * float(d["out_loss"][0])


def test_smplx_1024_frames_are_independent_and_deterministic():
    """BASELINE config 4 at full size: 1024 SMPL-X frames in one launch equal the same frames fitted in two halves, bit for
    bit (frames never interact), twice (determinism); the joint error drops."""
    from keypoints2body_amd import native, synthetic
    B = 1024
    p = synthetic.make_poses_x(B, seed=9)
    m, pr = H.native_model_x(), H.native_prior()
    pose = np.concatenate([getattr(p, k) for k, _ in POSE_FIELDS], axis=1)
    shape = np.concatenate([p.betas, p.expression], axis=1)
    j, _ = m.lbs(H.cuda(p.global_orient), H.cuda(pose), H.cuda(shape), H.cuda(p.transl), want_vertices=False)
    j3d = j[:, :55].contiguous()
    cfg = native.default_fit_config(); cfg.num_iters = 40; cfg.prior_pose_dims, cfg.num_betas_prior = 63, 10
    z = lambda c: torch.zeros(B, c, device="cuda")
    tr0 = (j3d[:, 0] - m.lbs(z(3), z(162), z(20), None, want_vertices=False)[0][:, 0]).contiguous()
    run = lambda sl: native.fit_world(m, pr, cfg, list(range(55)), j3d[sl].contiguous(), None, z(3)[sl], z(162)[sl], z(20)[sl], tr0[sl].contiguous())
    full, again = run(slice(0, B)), run(slice(0, B))
    a, b = run(slice(0, 500)), run(slice(500, B))
    for k in ("global_orient", "body_pose", "betas", "transl", "loss"):
        assert torch.equal(full[k], again[k]), k
        assert torch.equal(full[k], torch.cat([a[k], b[k]])), k
    jf, _ = m.lbs(full["global_orient"], full["body_pose"], full["betas"], full["transl"], want_vertices=False)
    err0 = (m.lbs(z(3), z(162), z(20), tr0, want_vertices=False)[0][:, :55] - j3d).norm(dim=-1).mean()
    err1 = (jf[:, :55] - j3d).norm(dim=-1).mean()
    assert err1 < 0.35 * err0, (float(err0), float(err1))


def test_smplx_through_the_public_sequence_api():
    """optimize_params_sequence with body_model="smplx" and a 55-joint model: SMPLXData results in both sequence modes;
    warm start = a chain of fit_frame calls, independent frames = rows of one batched fit (bit for bit: same launches)."""
    import keypoints2body_amd as k2b
    from keypoints2body_amd.core.fitters.world_space import WorldSpaceFitter
    from keypoints2body_amd.models.body_model import BodyModel
    from keypoints2body_amd.models.smpl_data import SMPLXData
    from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers
    d = H.load_smplx_case("amass22_zero_init")
    g = H.gmm_fixture()
    c = H.body_consts_x()
    model = BodyModel(c.v_template, c.shapedirs, c.posedirs, c.J_regressor, c.lbs_weights, c.parents, c.extra_vertex_ids)
    prior = MaxMixturePrior(MixtureBuffers(g["ref_means"], g["ref_precisions"], g["ref_nll_weights"].reshape(-1)))
    fields = ("global_orient", "body_pose", "transl", "left_hand_pose", "right_hand_pose", "expression", "jaw_pose", "leye_pose",
              "reye_pose", "betas")
    init = SMPLXData(**{k: torch.tensor(d["init_" + k][:1]) for k in fields})
    T = min(4, d["j3d"].shape[0])
    seq = d["j3d"][:T]
    mean = (torch.zeros(1, 66), torch.zeros(1, 10))
    fitter = WorldSpaceFitter(model, step_size=1e-2, num_iters_first=12, num_iters_followup=6, use_lbfgs=False,
                              joints_category="AMASS", pose_prior=prior)
    for warm in (True, False):
        cfg = {"frame": {"use_lbfgs": False, "num_iters_first": 12, "num_iters_followup": 6}, "use_shape_optimization": False,
               "use_previous_frame_init": warm, "fix_foot": False}
        res = k2b.optimize_params_sequence(seq, init_params=init, body_model="smplx", joint_layout="AMASS", model=model, config=cfg,
                                           pose_prior=prior, mean_params=mean)
        assert len(res) == T and all(isinstance(r.params, SMPLXData) for r in res)
        assert tuple(res[-1].vertices.shape) == (1, 10475, 3) and tuple(res[-1].joints.shape) == (1, 127, 3)
        prev = init
        for i in range(T):      # (warm start: the API runs the whole chain as ONE launch of the tree kernel - same bits as frame by frame)
            want = fitter.fit_frame(prev if warm or i == 0 else init, torch.tensor(seq[i:i + 1]), conf_3d=torch.ones(22), seq_ind=i)
            for k in fields:
                assert torch.equal(getattr(res[i].params, k), getattr(want.params, k)), (warm, i, k)
            assert torch.equal(res[i].vertices, want.vertices)
            prev = want.params
        # 22 AMASS targets give the fingers no gradient and no prior acts on them: they stay where they started
        assert float(res[-1].params.left_hand_pose.abs().max()) == 0.0 and float(res[-1].params.body_pose.abs().max()) > 0.0


@pytest.mark.parametrize("case,where", [("all55_zero_init", "out_"), ("all55_followup_frozen", "init_"), ("amass22_zero_init", "out_")])
def test_smplx_evaluate_only_gradient_matches_autograd(case, where):
    """What the tree kernel contributes to the L-BFGS branch: loss and analytic gradient at a given point (evaluate-only
    launch: one iteration, step size 0) against torch autograd through the oracle's SMPL-X forward and loss, at the
    reference's start / end points of the golden cases.  Gradient 5e-5 of its largest entry, loss 2e-5 relative."""
    from keypoints2body_amd import native
    from oracle.fit_torch import FitWeights, SMPLX_FIELDS, frame_losses
    d = H.load_smplx_case(case)
    seq_ind, freeze = int(d["seq_ind"]), bool(int(d["freeze_betas"]))
    idx = [int(i) for i in d["target_model_indices"]] if d["target_model_indices"].size else list(range(22))
    def autograd(dtype):
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=dtype)
        p = {k: t(d[where + k]).requires_grad_(True) for k in SMPLX_FIELDS}
        conf = t(d["conf"]) if int(d["has_conf"]) else torch.ones(len(idx), dtype=dtype)
        prior = H.oracle_prior()
        if dtype == torch.float64:
            import copy
            prior = copy.copy(prior)
            prior.means, prior.precisions, prior.nll_weights = prior.means.double(), prior.precisions.double(), prior.nll_weights.double()
        joints = H.oracle_model_x(double=dtype == torch.float64)(**p).joints
        lf = frame_losses(p["body_pose"], t(d["init_body_pose"]), p["betas"], joints[:, idx], t(d["j3d"]), prior, conf, FitWeights(),
                          preserve_on=seq_ind > 0)
        lf.sum().backward()
        g = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).double().numpy() for k, v in p.items()}
        if freeze:
            g["betas"][:] = 0.0
        return lf.detach().double().numpy(), np.concatenate([g["global_orient"]] + [g[k] for k, _ in POSE_FIELDS] +
                                                             [g["betas"], g["expression"], g["transl"]], axis=1)

    loss32, want32 = autograd(torch.float32)
    loss64, want = autograd(torch.float64)

    cfg = native.default_fit_config()
    cfg.num_iters, cfg.step_size = 1, 0.0
    cfg.pose_preserve_weight = 5.0 if seq_ind > 0 else 0.0
    cfg.freeze_betas = int(freeze)
    cfg.prior_pose_dims, cfg.num_betas_prior = 63, 10
    go, pose, shape, tr = map(H.cuda, pack(d, where))
    pres = H.cuda(np.concatenate([d["init_" + k] for k, _ in POSE_FIELDS], axis=1))
    out = native.fit_world(H.native_model_x(), H.native_prior(), cfg, idx, H.cuda(d["j3d"]), H.cuda(d["conf"]) if int(d["has_conf"]) else None,
                           go, pose, shape, tr, preserve_pose=pres, want_grad=True)
    np.testing.assert_allclose(out["loss"].cpu().numpy(), loss64, rtol=2e-5)
    got = out["grad"].cpu().numpy().astype(np.float64)
    scale = np.abs(want).max(axis=1, keepdims=True)
    err = (np.abs(got - want) / scale).max()
    err32 = (np.abs(want32 - want) / scale).max()          # what fp32 autograd itself loses at this point (near a minimum the
                                                           # gradient is a small difference of large terms)
    assert err < max(5e-5, 2.0 * err32), f"{case}/{where}: gradient off by {err:.2e} of its largest entry (fp32 autograd: {err32:.2e})"
    print(f"smplx gradient {case}/{where}: HIP {err:.2e}, fp32 autograd {err32:.2e} of the largest entry (float64 reference)")
    for k in ("global_orient", "body_pose", "betas", "transl"):          # an evaluate-only launch moves nothing
        assert torch.equal(out[k], {"global_orient": go, "body_pose": pose, "betas": shape, "transl": tr}[k])


def test_smplx_lbfgs_branch_runs_on_evaluate_only_launches():
    """use_lbfgs=True (the reference default, world_space.py:231-247) with a 55-joint model: torch.optim.LBFGS over the packed
    parameter vectors, every closure call one evaluate-only launch of the tree kernel.  No golden (the reference cannot run
    SMPL-X with its prior, SURVEY N3, and the mode is chaotic): it must beat the 30 Adam steps' loss from the same
    start, keep frozen betas bit for bit and leave the untargeted fingers alone."""
    from keypoints2body_amd.core.fitters.world_space import WorldSpaceFitter
    from keypoints2body_amd.models.body_model import BodyModel
    from keypoints2body_amd.models.smpl_data import SMPLXData
    from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers
    d = H.load_smplx_case("amass22_zero_init")
    g = H.gmm_fixture()
    c = H.body_consts_x()
    model = BodyModel(c.v_template, c.shapedirs, c.posedirs, c.J_regressor, c.lbs_weights, c.parents, c.extra_vertex_ids)
    prior = MaxMixturePrior(MixtureBuffers(g["ref_means"], g["ref_precisions"], g["ref_nll_weights"].reshape(-1)))
    fields = ("global_orient", "body_pose", "transl", "left_hand_pose", "right_hand_pose", "expression", "jaw_pose", "leye_pose",
              "reye_pose", "betas")
    init = SMPLXData(**{k: torch.tensor(d["init_" + k][:2]) for k in fields})
    j3d = torch.tensor(d["j3d"][:2])
    kw = dict(step_size=1e-2, num_iters_first=30, num_iters_followup=30, joints_category="AMASS", pose_prior=prior)
    adam = WorldSpaceFitter(model, use_lbfgs=False, **kw).fit_frame(init, j3d, seq_ind=0)
    for freeze in (False, True):
        res = WorldSpaceFitter(model, use_lbfgs=True, **kw).fit_frame(init, j3d, seq_ind=1 if freeze else 0, freeze_betas=freeze)
        assert isinstance(res.params, SMPLXData) and torch.isfinite(res.loss)
        assert all(torch.isfinite(getattr(res.params, k)).all() for k in fields)
        if freeze:
            assert torch.equal(res.params.betas.cpu(), init.betas)
        else:
            assert float(res.loss) < float(adam.loss)
        assert float(res.params.left_hand_pose.abs().max()) == 0.0


def test_smplx_chain_launch_equals_frame_by_frame_launches():
    """k2b_fit_sequence on the 55-joint tree kernel: many chains side by side, per-frame confidences, frozen betas, follow-up
    count larger than the first frame's - bit-identical to host-driven per-frame launches."""
    from keypoints2body_amd import native, synthetic
    import copy
    m, pr = H.native_model_x(), H.native_prior()
    for S, T, first, follow, freeze in ((1, 5, 12, 6, False), (37, 3, 5, 9, True)):
        p = synthetic.make_poses_x(S * T, seed=S)
        pose = np.concatenate([getattr(p, k) for k, _ in POSE_FIELDS], axis=1)
        shape = np.concatenate([p.betas, p.expression], axis=1)
        j, _ = m.lbs(H.cuda(p.global_orient), H.cuda(pose), H.cuda(shape), H.cuda(p.transl), want_vertices=False)
        j3d = j[:, :55].reshape(S, T, 55, 3).contiguous()
        conf = H.cuda(np.random.default_rng(S).uniform(0.5, 1.5, (S, T, 55)).astype(np.float32))
        z = lambda c: torch.zeros(S, c, device="cuda")
        go, bp, be, tr = z(3), z(162), 0.1 * torch.ones(S, 20, device="cuda"), j3d[:, 0, 0].contiguous()
        cfg = native.default_fit_config()
        cfg.num_iters, cfg.pose_preserve_weight, cfg.freeze_betas, cfg.conf_per_frame = first, 5.0, int(freeze), 1
        cfg.prior_pose_dims, cfg.num_betas_prior = 63, 10
        got = native.fit_sequence(m, pr, cfg, follow, list(range(55)), j3d, conf, go, bp, be, tr)
        cur, want = (go, bp, be, tr), {k: [] for k in got}
        for t in range(T):
            c = copy.copy(cfg)
            c.num_iters, c.pose_preserve_weight = (first, 0.0) if t == 0 else (follow, 5.0)
            o = native.fit_world(m, pr, c, list(range(55)), j3d[:, t].contiguous(), conf[:, t].contiguous(), *cur)
            for k in want:
                want[k].append(o[k])
            cur = (o["global_orient"], o["body_pose"], o["betas"], o["transl"])
        for k in got:
            assert torch.equal(got[k], torch.stack(want[k], dim=1)), (S, T, k)
        if freeze:
            assert torch.equal(got["betas"][:, :, :10], be[:, None, :10].expand(-1, T, -1))


def test_smplx_fit_does_not_depend_on_the_workgroup_shape():
    """The tree kernel runs 1..8 frames (waves) per workgroup by batch size, and the mixture's eight components are spread over
    those waves (1, 2, 3, 5 or 8 waves: even and uneven shares): the same frame must come out bit for bit whatever batch it
    rides in, whichever MFMA column it occupies."""
    from keypoints2body_amd import native, synthetic
    m, pr = H.native_model_x(), H.native_prior()
    B = 2048
    p = synthetic.make_poses_x(B, seed=11)
    pose = np.concatenate([getattr(p, k) for k, _ in POSE_FIELDS], axis=1)
    shape = np.concatenate([p.betas, p.expression], axis=1)
    j, _ = m.lbs(H.cuda(p.global_orient), H.cuda(pose), H.cuda(shape), H.cuda(p.transl), want_vertices=False)
    j3d = (j[:, :55] + 0.01).contiguous()
    cfg = native.default_fit_config(); cfg.num_iters = 15; cfg.prior_pose_dims, cfg.num_betas_prior = 63, 10
    z = lambda n, c: torch.zeros(n, c, device="cuda")
    run = lambda n: native.fit_world(m, pr, cfg, list(range(55)), j3d[:n].contiguous(), None, z(n, 3), z(n, 162), z(n, 20),
                                     j3d[:n, 0].contiguous())
    ref = run(B)                                   # 8 waves per workgroup
    for n in (1, 300, 700, 1200):                  # 1, 2, 3 and 5 waves per workgroup on a 256-CU device
        out = run(n)
        for k in ("global_orient", "body_pose", "betas", "transl", "loss"):
            assert torch.equal(out[k], ref[k][:n]), (n, k)


def test_smplx_vertex_term_gradient_matches_autograd():
    """k2b_vertex_term on the 55-joint tree (the 64-joint / 32-coefficient instantiation of the kernel): loss and analytic
    gradient of the joint loss on vertex-selected joints against torch autograd through the oracle's SMPL-X forward."""
    from keypoints2body_amd import native, synthetic
    from oracle.fit_torch import SMPLX_FIELDS, gmof
    B = 3
    sel = [0, 6, 20, 45, 71]
    p = synthetic.make_poses_x(B, seed=21)
    t = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32)
    q = {k: t(getattr(p, k)).requires_grad_() for k in SMPLX_FIELDS}
    joints = H.oracle_model_x()(**q).joints
    gen = torch.Generator().manual_seed(3)
    tgt = joints[:, [55 + e for e in sel]].detach() + 0.05 * torch.randn(B, len(sel), 3, generator=gen)
    conf = torch.tensor([1.0, 0.7, 1.5, 1.0, 0.9])
    lf = ((600.0 ** 2) * (conf ** 2).view(1, -1, 1) * gmof(joints[:, [55 + e for e in sel]] - tgt, 100.0)).sum(dim=(1, 2))
    lf.sum().backward()
    g_ref = torch.cat([q["global_orient"].grad] + [q[k].grad for k, _ in POSE_FIELDS] + [q["betas"].grad, q["expression"].grad,
                       q["transl"].grad], dim=1).numpy()
    pose = np.concatenate([getattr(p, k) for k, _ in POSE_FIELDS], axis=1)
    shape = np.concatenate([p.betas, p.expression], axis=1)
    loss, grad = native.vertex_term(H.native_model_x(), sel, tgt.cuda().contiguous(), conf.cuda(), 100.0, 600.0,
                                    H.cuda(p.global_orient), H.cuda(pose), H.cuda(shape), H.cuda(p.transl))
    np.testing.assert_allclose(loss.cpu().numpy(), lf.detach().numpy(), rtol=2e-5)
    g = grad.cpu().numpy()
    for name, sl in (("global_orient", slice(0, 3)), ("pose", slice(3, 165)), ("shape", slice(165, 185)), ("transl", slice(185, 188))):
        scale = np.abs(g_ref[:, sl]).max()
        assert np.abs(g[:, sl] - g_ref[:, sl]).max() / scale < 5e-5, name


def test_smplx_vertex_joints_with_frozen_betas_and_per_frame_confidences():
    """Vertex-selected joints on the 55-joint tree through k2b_fit_world: frozen betas stay put bit for bit while the expression
    moves (the Adam tail's per-coefficient mask), and a (B, K) confidence tensor gives row by row the single-frame results."""
    from keypoints2body_amd import native
    d = H.load_smplx_case("vertex_joints_zero_init")
    idx = [int(i) for i in d["target_model_indices"]]
    go, pose, shape, tr = map(H.cuda, pack(d, "init_"))
    shape = shape + 0.05
    B, K = d["j3d"].shape[:2]
    conf = H.cuda(np.random.default_rng(1).uniform(0.5, 1.5, (B, K)).astype(np.float32))
    cfg = native.default_fit_config()
    cfg.num_iters, cfg.freeze_betas, cfg.conf_per_frame = 10, 1, 1
    cfg.prior_pose_dims, cfg.num_betas_prior = 63, 10
    out = native.fit_world(H.native_model_x(), H.native_prior(), cfg, idx, H.cuda(d["j3d"]), conf, go, pose, shape, tr)
    assert torch.equal(out["betas"][:, :10], shape[:, :10]) and not torch.equal(out["betas"][:, 10:], shape[:, 10:])
    cfg.conf_per_frame = 0
    for f in range(B):
        sl = slice(f, f + 1)
        one = native.fit_world(H.native_model_x(), H.native_prior(), cfg, idx, H.cuda(d["j3d"][sl]), conf[f].contiguous(), go[sl].contiguous(),
                               pose[sl].contiguous(), shape[sl].contiguous(), tr[sl].contiguous())
        for k in ("global_orient", "body_pose", "betas", "transl", "loss"):
            assert torch.equal(out[k][sl], one[k]), (f, k)


def test_smplx_sequence_with_the_reference_default_configuration():
    """optimize_params_sequence with its DEFAULTS (shape pre-pass on, L-BFGS per frame, warm start) and a 55-joint model: the
    pre-pass fits the 10 betas with the expression at zero, every frame comes back as SMPLXData with finite values, and the fit
    explains the targets."""
    import keypoints2body_amd as k2b
    from keypoints2body_amd.models.body_model import BodyModel
    from keypoints2body_amd.models.smpl_data import SMPLXData
    from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers
    d = H.load_smplx_case("amass22_zero_init")
    g = H.gmm_fixture()
    prior = MaxMixturePrior(MixtureBuffers(g["ref_means"], g["ref_precisions"], g["ref_nll_weights"].reshape(-1)))
    model = BodyModel.synthetic_x(0)
    seq = np.tile(d["j3d"], (3, 1, 1))[:4]
    res = k2b.optimize_params_sequence(seq, body_model="smplx", joint_layout="AMASS", model=model, pose_prior=prior,
                                       mean_params=(torch.zeros(1, 66), torch.zeros(1, 10)))
    assert len(res) == 4 and all(isinstance(r.params, SMPLXData) for r in res)
    for r in res:
        assert torch.isfinite(r.loss) and all(torch.isfinite(getattr(r.params, k)).all() for k in ("betas", "body_pose", "expression", "transl"))
    assert float(res[0].params.betas.abs().max()) > 1e-3          # the pre-pass moved the shape off the zero mean
    with torch.no_grad():                                          # the zero pose, root aligned: where the fit starts
        j0 = model(global_orient=torch.zeros(1, 3), body_pose=torch.zeros(1, 63), return_verts=False).joints[:, :22].cpu()
    tgt = torch.tensor(seq[-1:])
    err0 = (j0 - j0[:, :1] + tgt[:, :1] - tgt).norm(dim=-1).mean()
    err = (res[-1].joints[:, :22].cpu() - tgt).norm(dim=-1).mean()
    assert float(err) < 0.6 * float(err0), (float(err0), float(err))   # (30 + 10 L-BFGS iterations: the reference's defaults)


def test_prior_width_can_change_between_calls_on_one_model():
    """The lane table's prior columns follow k2b_fit_config.prior_pose_dims from call to call (state kept in the model handle)."""
    from keypoints2body_amd import native
    d = H.load_smplx_case("all55_zero_init")
    idx = [int(i) for i in d["target_model_indices"]]
    go, pose, shape, tr = map(H.cuda, pack(d, "init_"))

    def run(dims):
        cfg = native.default_fit_config()
        cfg.num_iters, cfg.prior_pose_dims, cfg.num_betas_prior = 8, dims, 10
        return native.fit_world(H.native_model_x(), H.native_prior(), cfg, idx, H.cuda(d["j3d"]), H.cuda(d["conf"]), go, pose, shape, tr)

    a, b, c = run(63), run(60), run(63)
    assert torch.equal(a["body_pose"], c["body_pose"]) and torch.equal(a["loss"], c["loss"])
    assert not torch.equal(a["body_pose"], b["body_pose"])


@pytest.mark.parametrize("seed", list(range(1, 9)))
def test_smplx_random_configurations_match_oracle(seed):
    """Seeded sweep over the configuration space of the tree kernel against the oracle (itself pinned to the reference on the
    golden cases): batch size (1-9 frames: 1-8 waves per workgroup, uneven component shares), iteration count, target subset and
    order (body, face, fingers), confidences incl. zeros and > 1, every loss weight (some zero: no mixture at all), sigma,
    frozen betas, first / follow-up frame, zero / perturbed / large starts.  Parameters within 1e-4, last loss 2e-4 relative."""
    from keypoints2body_amd import native, synthetic
    from oracle.fit_torch import FitWeights, SMPLX_FIELDS, fit_world_adam_smplx
    rng = np.random.default_rng(1000 + seed)
    B = int(rng.integers(1, 10))
    K = int(rng.integers(8, 56))
    idx = sorted(rng.choice(55, size=K, replace=False).tolist())
    idx = [idx[i] for i in rng.permutation(K)]
    if 0 not in idx:
        idx[0] = 0
    p = synthetic.make_poses_x(B, seed=200 + seed)
    t = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32)
    truth = {k: t(getattr(p, k)) for k in SMPLX_FIELDS}
    kind = ("zeros", "perturbed", "large")[seed % 3]
    if kind == "zeros":
        init = {k: torch.zeros_like(v) for k, v in truth.items()}
    elif kind == "perturbed":
        init = {k: v * 0.7 + 0.02 for k, v in truth.items()}
        init["body_pose"][:, 6:12] = 0.0
    else:
        init = {k: v * 1.5 for k, v in truth.items()}
        axis = t(rng.standard_normal((B, 3))); init["global_orient"] = axis / axis.norm(dim=1, keepdim=True) * 3.1
        init["betas"] = t(rng.uniform(-2, 2, (B, 10)))
    oracle = H.oracle_model_x()
    with torch.no_grad():
        j3d = oracle(**truth).joints[:, idx] + t(0.01 * rng.standard_normal((B, K, 3)))
        j0 = oracle(**{k: v for k, v in init.items() if k != "transl"}).joints
    init["transl"] = (j3d[:, idx.index(0)] - j0[:, 0]) + 0.01
    conf = t(rng.choice([0.0, 0.5, 1.0, 1.5], size=K, p=[0.15, 0.2, 0.45, 0.2])); conf[idx.index(0)] = 1.0
    pick = lambda *v: float(rng.choice(v))
    weights = dict(sigma=pick(30.0, 100.0, 300.0), pose_prior_weight=pick(0.0, 4.78 * 1.5, 12.0), shape_prior_weight=pick(0.0, 5.0, 20.0),
                   angle_prior_weight=pick(0.0, 15.2), joint_loss_weight=pick(100.0, 600.0), pose_preserve_weight=pick(1.0, 5.0))
    iters, seq_ind, freeze = int(rng.integers(3, 26)), int(seed % 2) * 3, bool(seed % 4 == 0)
    ref, ref_loss, _, _, _ = fit_world_adam_smplx(oracle, H.oracle_prior(), init, j3d, conf if seed % 3 else None, num_iters=iters,
                                                  seq_ind=seq_ind, model_idx=idx, weights=FitWeights(**weights), freeze_betas=freeze)
    cfg = native.default_fit_config()
    cfg.num_iters = iters
    for k, v in weights.items():
        setattr(cfg, k, v)
    if seq_ind == 0:
        cfg.pose_preserve_weight = 0.0
    cfg.freeze_betas, cfg.prior_pose_dims, cfg.num_betas_prior = int(freeze), 63, 10
    pose0 = torch.cat([init[k] for k, _ in POSE_FIELDS], dim=1)
    shape0 = torch.cat([init["betas"], init["expression"]], dim=1)
    out = native.fit_world(H.native_model_x(), H.native_prior(), cfg, idx, j3d.cuda().contiguous(), conf.cuda() if seed % 3 else None,
                           init["global_orient"].cuda().contiguous(), pose0.cuda().contiguous(), shape0.cuda().contiguous(),
                           init["transl"].cuda().contiguous())
    want_pose = torch.cat([ref[k] for k, _ in POSE_FIELDS], dim=1)
    want_shape = torch.cat([ref["betas"], ref["expression"]], dim=1)
    tag = (seed, kind, B, iters, K, weights["pose_prior_weight"])
    worst = 0.0
    for key, want in (("global_orient", ref["global_orient"]), ("body_pose", want_pose), ("betas", want_shape), ("transl", ref["transl"])):
        err = (out[key].cpu() - want).abs().max().item()
        worst = max(worst, err)
        assert err < PARAM_TOL, (tag, key, err)
    np.testing.assert_allclose(out["loss"].cpu().numpy(), ref_loss.numpy(), rtol=2e-4, err_msg=str(tag))
    if freeze:
        assert torch.equal(out["betas"][:, :10].cpu(), init["betas"])
    print(f"smplx random configuration {tag}: worst parameter deviation {worst:.2e}")
