"""GPU: the warm-start chain of the reference's sequence mode (``api/sequence.py:214-281`` with
``use_previous_frame_init=True``) as ONE launch (``k2b_fit_sequence``).

Gates:
* against the oracle through the public API: ``test_gpu_api.py::test_sequence_warm_start_chain_matches_oracle``
  (that test now runs this path);
* here: the in-kernel frame loop against the same engine driven frame by frame from the host (one
  ``k2b_fit_world`` launch per frame, each frame starting from its predecessor's downloaded result) - the same
  arithmetic in the same launch shape, so the results must agree to the last bit; many chains side by side
  (several sequences per workgroup, a padding slot), per-frame confidences, frozen betas, a follow-up count
  larger than the first frame's (the Adam table's prefix rule), and a chain of one.
"""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def stepwise(cfg, followup, idx, j3d, conf, go, bp, be, tr):
    """Host-driven chain: T launches of k2b_fit_world over the S sequences (the reference's frame loop)."""
    from keypoints2body_amd import native
    import copy
    S, T = j3d.shape[:2]
    outs = {k: [] for k in ("global_orient", "body_pose", "betas", "transl", "loss")}
    cur = (go, bp, be, tr)
    first_iters, w = int(cfg.num_iters), float(cfg.pose_preserve_weight)
    for t in range(T):
        c = copy.copy(cfg)
        c.num_iters = first_iters if t == 0 else followup
        c.pose_preserve_weight = 0.0 if t == 0 else w
        c.debug_launch_shape = H.LAUNCH_SHAPES["split"]
        cf = conf if (conf is None or conf.dim() == 1) else conf[:, t].contiguous()
        o = native.fit_world(H.native_model(), H.native_prior(), c, idx, j3d[:, t].contiguous(), cf, *cur)
        for k in outs:
            outs[k].append(o[k])
        cur = (o["global_orient"], o["body_pose"], o["betas"], o["transl"])
    return {k: torch.stack(v, dim=1) for k, v in outs.items()}


def problem(S, T, seed=0, per_frame_conf=False):
    from keypoints2body_amd import synthetic
    rng = np.random.default_rng(seed)
    poses = synthetic.make_poses(S * T, seed=seed + 1)
    model = H.oracle_model()
    with torch.no_grad():
        t = lambda a: torch.tensor(np.asarray(a, np.float32))
        j = model(global_orient=t(poses.global_orient), body_pose=t(poses.body_pose), betas=t(poses.betas),
                  transl=t(poses.transl)).joints[:, :22]
    j3d = (j.numpy() + rng.normal(0, 0.01, j.shape)).astype(np.float32).reshape(S, T, 22, 3)
    conf = rng.uniform(0.5, 1.5, (S, T, 22) if per_frame_conf else (22,)).astype(np.float32)
    go = rng.normal(0, 0.2, (S, 3)).astype(np.float32)
    bp = rng.normal(0, 0.1, (S, 69)).astype(np.float32)
    be = rng.normal(0, 0.5, (S, 10)).astype(np.float32)
    tr = (j3d[:, 0, 0] + rng.normal(0, 0.05, (S, 3))).astype(np.float32)
    return [H.cuda(x) for x in (j3d, conf, go, bp, be, tr)]


@pytest.mark.parametrize("S,T,first,follow,per_frame,freeze", [
    (1, 6, 30, 10, False, False),        # the API's case: one sequence
    (5, 4, 12, 5, True, False),          # per-frame confidences
    (301, 3, 8, 12, False, True),        # two sequences per workgroup + a padding slot; follow-up count > first
    (1100, 2, 5, 3, True, False),        # more chains than one slot per CU: four per workgroup
])
def test_chain_launch_equals_frame_by_frame_launches(S, T, first, follow, per_frame, freeze):
    from keypoints2body_amd import native
    j3d, conf, go, bp, be, tr = problem(S, T, seed=S, per_frame_conf=per_frame)
    cfg = native.default_fit_config()
    cfg.num_iters, cfg.pose_preserve_weight, cfg.freeze_betas, cfg.conf_per_frame = first, 5.0, int(freeze), int(per_frame)
    idx = list(range(22))
    got = native.fit_sequence(H.native_model(), H.native_prior(), cfg, follow, idx, j3d, conf, go, bp, be, tr)
    want = stepwise(cfg, follow, idx, j3d, conf, go, bp, be, tr)
    for k in want:
        assert torch.equal(got[k], want[k]), (k, (got[k] - want[k]).abs().max().item())
    if freeze:
        assert torch.equal(got["betas"], be[:, None].expand(-1, T, -1))


def test_chain_of_one_is_the_first_frame_fit():
    from keypoints2body_amd import native
    j3d, conf, go, bp, be, tr = problem(7, 1, seed=3)
    cfg = native.default_fit_config()
    cfg.num_iters, cfg.pose_preserve_weight = 9, 5.0
    got = native.fit_sequence(H.native_model(), H.native_prior(), cfg, 4, list(range(22)), j3d, conf, go, bp, be, tr)
    cfg.pose_preserve_weight = 0.0
    want = native.fit_world(H.native_model(), H.native_prior(), cfg, list(range(22)), j3d[:, 0].contiguous(), conf, go, bp, be, tr)
    for k in want:
        assert torch.equal(got[k][:, 0], want[k]), k


def test_chain_rejects_what_it_does_not_cover():
    from keypoints2body_amd import native
    j3d, conf, go, bp, be, tr = problem(2, 3, seed=5)
    cfg = native.default_fit_config()
    cfg.transl_prior_weight = 1.0
    with pytest.raises(RuntimeError):
        native.fit_sequence(H.native_model(), H.native_prior(), cfg, 4, list(range(22)), j3d, conf, go, bp, be, tr)
    cfg = native.default_fit_config()
    with pytest.raises(ValueError):
        native.fit_sequence(H.native_model(), H.native_prior(), cfg, 0, list(range(22)), j3d, conf, go, bp, be, tr)


def test_fitter_chain_through_the_sequence_api_runs_one_fit_launch():
    """optimize_params_sequence (warm start, Adam, SMPL) makes ONE fit call for the whole sequence."""
    import keypoints2body_amd as k2b
    from keypoints2body_amd import native
    from keypoints2body_amd.models.body_model import BodyModel
    from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers
    g = H.gmm_fixture()
    prior = MaxMixturePrior(MixtureBuffers(g["ref_means"], g["ref_precisions"], g["ref_nll_weights"].reshape(-1)))
    model = BodyModel.synthetic(0)
    d = H.load_case("amass_noisy_conf")
    calls = {"seq": 0, "world": 0}
    orig_s, orig_w = native.fit_sequence, native.fit_world
    def spy_s(*a, **k):
        calls["seq"] += 1
        return orig_s(*a, **k)
    def spy_w(*a, **k):
        calls["world"] += 1
        return orig_w(*a, **k)
    native.fit_sequence, native.fit_world = spy_s, spy_w
    try:
        pose = torch.tensor(np.concatenate([d["init_global_orient"][:1], d["init_body_pose"][:1]], axis=1))
        res = k2b.optimize_params_sequence(d["j3d"][:4], model=model, pose_prior=prior,
                                           config={"frame": {"use_lbfgs": False}, "use_shape_optimization": False},
                                           mean_params=(pose, torch.tensor(d["init_betas"][:1])))
    finally:
        native.fit_sequence, native.fit_world = orig_s, orig_w
    assert len(res) == 4 and calls == {"seq": 1, "world": 0}
    assert all(tuple(r.vertices.shape) == (1, 6890, 3) for r in res)


@pytest.mark.parametrize("name", H.CHAIN_CASES)
def test_chain_launch_matches_reference_sequence_loop_on_real_motion(name):
    """``k2b_fit_sequence`` against the REAL reference walked through its own frame loop (api/sequence.py:124-128, 214-281:
    fix_foot confidences per frame, seq_ind = idx, prev = res.params) over the first frames of the reference's demo
    motions (``tests/golden/chain_motion*.npz`` from oracle/gen_golden_chain.py).

    FREE-RUNNING (the one-launch chain): a chain amplifies rounding - the reference walked again from a start perturbed by
    2e-6 relative is up to 2.6e-2 away from itself after 14 frames (``out_param_dev_perturbed``) - so frame t may deviate by
    max(1e-4, 4 x the reference's own deviation at that frame); the first frames, where that floor is ~1e-5, are held to 1e-4.
    TEACHER-FORCED (every frame fitted from the REFERENCE's result of its predecessor, all frames in one batched launch
    with per-frame confidences): 1e-4 on EVERY frame for parameters, joints and sampled vertices, 1e-4 relative for the
    loss - the loop's per-frame semantics without accumulation.  (The chain launch itself is bit-identical to such
    frame-by-frame launches: the tests above.)"""
    from keypoints2body_amd import native
    d = H.load_chain_case(name)
    T = d["j3d"].shape[0]
    keys = ("global_orient", "body_pose", "betas", "transl")
    cfg = native.default_fit_config()
    cfg.num_iters = int(d["num_iters_first"])
    cfg.pose_preserve_weight = 5.0
    cfg.conf_per_frame = 1
    out = native.fit_sequence(H.native_model(), H.native_prior(), cfg, int(d["num_iters_followup"]), list(range(22)),
                              H.cuda(d["j3d"][None]), H.cuda(d["conf"][None]), H.cuda(d["init_global_orient"]),
                              H.cuda(d["init_body_pose"]), H.cuda(d["init_betas"]), H.cuda(d["init_transl"]))
    dev = torch.stack([(out[k][0].cpu() - torch.tensor(d["out_" + k])).abs().amax(dim=1) for k in keys]).amax(dim=0).numpy()
    floor = np.maximum(1e-4, 4.0 * d["out_param_dev_perturbed"])
    assert np.all(dev < floor), (dev.tolist(), floor.tolist())
    assert np.all(dev[:4] < 1e-4)
    # teacher-forced: frame 0 from the start, frames 1.. from the reference's previous result (preserve = that start)
    first = {k: out[k][0, :1] for k in keys + ("loss",)}
    cfg2 = native.default_fit_config()
    cfg2.num_iters = int(d["num_iters_followup"])
    cfg2.pose_preserve_weight = 5.0
    cfg2.conf_per_frame = 1
    rest = native.fit_world(H.native_model(), H.native_prior(), cfg2, list(range(22)), H.cuda(d["j3d"][1:]), H.cuda(d["conf"][1:]),
                            *[H.cuda(d["out_" + k][:-1]) for k in keys])
    tf = {k: torch.cat([first[k], rest[k]]) for k in keys + ("loss",)}
    worst = 0.0
    for k in keys:
        e = (tf[k].cpu() - torch.tensor(d["out_" + k])).abs().amax(dim=1)
        worst = max(worst, float(e.max()))
        assert float(e.max()) < 1e-4, (k, e.tolist())
    rel = ((tf["loss"].cpu() - torch.tensor(d["out_loss"])).abs() / torch.tensor(d["out_loss"]).abs()).max()
    assert float(rel) < 1e-4, float(rel)
    joints, verts = H.native_model().lbs(*[tf[k] for k in keys])
    assert float((joints.cpu() - torch.tensor(d["out_joints"])).abs().max()) < 1e-4
    vid = torch.tensor(d["sampled_vertex_ids"])
    assert float((verts.cpu()[:, vid] - torch.tensor(d["out_verts_sampled"])).abs().max()) < 1e-4
    print(f"{name}: {T} frames; teacher-forced worst {worst:.2e}; free-running per frame {np.array2string(dev, precision=1)}")


def test_public_sequence_api_reproduces_reference_loop_on_real_motion():
    """The same golden through ``optimize_params_sequence`` (fix_foot on, warm start, Adam, defaults 30 / 10): the public
    function must apply the foot confidences itself and walk the chain as the reference's loop does."""
    import keypoints2body_amd as k2b
    from keypoints2body_amd.models.body_model import BodyModel
    from keypoints2body_amd.models.smpl_data import SMPLData
    from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers
    d = H.load_chain_case("chain_motion1_30_10")
    g = H.gmm_fixture()
    prior = MaxMixturePrior(MixtureBuffers(g["ref_means"], g["ref_precisions"], g["ref_nll_weights"].reshape(-1)))
    t = lambda k: torch.tensor(d[k])
    init = SMPLData(betas=t("init_betas"), global_orient=t("init_global_orient"), body_pose=t("init_body_pose"),
                    transl=t("init_transl"))
    cfg = {"frame": {"use_lbfgs": False, "num_iters_first": 30, "num_iters_followup": 10}, "use_shape_optimization": False,
           "use_previous_frame_init": True, "fix_foot": True}
    res = k2b.optimize_params_sequence(d["j3d"], init_params=init, body_model="smpl", joint_layout="AMASS",
                                       model=BodyModel.synthetic(0), config=cfg, pose_prior=prior,
                                       mean_params=(torch.zeros(1, 72), torch.zeros(1, 10)))
    assert len(res) == d["j3d"].shape[0]
    for i, r in enumerate(res):                      # free-running chain: the reference's own noise floor applies (see above)
        floor = max(1e-4, 4.0 * float(d["out_param_dev_perturbed"][i]))
        for k in ("global_orient", "body_pose", "betas", "transl"):
            assert float((getattr(r.params, k).cpu() - t("out_" + k)[i:i + 1]).abs().max()) < floor, (i, k)
        assert float((r.joints.cpu() - t("out_joints")[i:i + 1]).abs().max()) < 10 * floor, i
