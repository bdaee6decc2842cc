"""GPU: size-independent properties at BASELINE.json's full sizes (1024 / 4096 / 10 000 frames) and
the edge cases (empty, single, ragged batches).  No oracle at these sizes: the properties are exact
consequences of the path's structure (frames are independent fits; the loss is invariant under a
common translation of targets and transl; at zero pose LBS is the shape blend)."""
import numpy as np
import pytest
import torch

from keypoints2body_amd import native, synthetic
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _problem(B, seed=11):
    m = H.native_model()
    p = synthetic.make_poses(B, seed=seed)
    j, _ = m.lbs(*map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl)), want_vertices=False)
    j3d = j[:, :22].contiguous()
    z = lambda c: torch.zeros(B, c, device="cuda")
    j0, _ = m.lbs(z(3), z(69), z(10), None, want_vertices=False)
    return j3d, (j3d[:, 0] - j0[:, 0]).contiguous()


def _fit(j3d, tr0, iters=100, rows=None):
    B = j3d.shape[0]
    cfg = native.default_fit_config()
    cfg.num_iters = iters
    z = lambda c: torch.zeros(B, c, device="cuda")
    return native.fit_world(H.native_model(), H.native_prior(), cfg, list(range(22)), j3d, None, z(3), z(69), z(10), tr0)


@pytest.mark.parametrize("B", [1024, 2048, 4096, 8999, 10000])      # split, split-paired, paired, whole rounds + a remainder launch (split / split-paired)
def test_full_size_fit_is_deterministic_independent_and_converges(B):
    j3d, tr0 = _problem(B)
    a, b = _fit(j3d, tr0), _fit(j3d, tr0)
    for k in ("global_orient", "body_pose", "betas", "transl", "loss"):
        assert torch.equal(a[k], b[k]), k                       # run-to-run bit-exact
        assert torch.isfinite(a[k]).all()
    # frames are independent fits: any sub-batch reproduces its rows bit for bit, whatever launch shape
    # (launch shape, workgroup size) the batch size selects
    for sl in (slice(0, 1), slice(B // 2 - 3, B // 2 + 4), slice(B - 700, B)):
        sub = _fit(j3d[sl].contiguous(), tr0[sl].contiguous())
        for k in ("global_orient", "body_pose", "betas", "transl"):
            assert (sub[k] - a[k][sl]).abs().max() < 2e-5, (k, sl)   # different launch shapes may differ in rounding only
    joints, verts = H.native_model().lbs(a["global_orient"], a["body_pose"], a["betas"], a["transl"])
    err = (joints[:, :22] - j3d).norm(dim=-1).mean().item()
    assert err < 0.03                                            # 100 Adam iterations from zero pose: ~2 cm
    assert verts.shape == (B, 6890, 3) and torch.isfinite(verts).all()


@pytest.mark.parametrize("B", [3, 1500, 2600])        # split (ragged), split-paired, paired
def test_long_runs_stay_finite_and_keep_improving(B):
    """1000 Adam iterations in one launch (ten times the benchmark's count): the in-kernel wave
    synchronisation (barriers and the row waves' LDS counter) holds up, parameters stay finite and the joint
    error does not get worse than after 100 iterations."""
    j3d, tr0 = _problem(B)
    short, long_ = _fit(j3d, tr0, iters=100), _fit(j3d, tr0, iters=1000)
    model = H.native_model()
    err = {}
    for name, out in (("short", short), ("long", long_)):
        for k in ("global_orient", "body_pose", "betas", "transl", "loss"):
            assert torch.isfinite(out[k]).all(), (name, k)
        joints, _ = model.lbs(out["global_orient"], out["body_pose"], out["betas"], out["transl"], want_vertices=False)
        err[name] = (joints[:, :22] - j3d).norm(dim=-1).mean().item()
    assert err["long"] <= err["short"] * 1.05, err


def test_translation_equivariance_of_the_fit():
    j3d, tr0 = _problem(512, seed=3)
    shift = torch.tensor([0.25, -1.5, 3.0], device="cuda")
    a = _fit(j3d, tr0, iters=60)
    b = _fit((j3d + shift).contiguous(), (tr0 + shift).contiguous(), iters=60)
    assert (b["transl"] - a["transl"] - shift).abs().max() < 5e-5
    for k in ("global_orient", "body_pose", "betas"):
        assert (b[k] - a[k]).abs().max() < 5e-5, k


@pytest.mark.parametrize("B", [0, 1, 5, 33, 257])
def test_empty_single_and_ragged_batches(B):
    m = H.native_model()
    if B == 0:
        e = lambda c: torch.zeros(0, c, device="cuda")
        out = native.fit_world(m, H.native_prior(), native.default_fit_config(), list(range(22)),
                               torch.zeros(0, 22, 3, device="cuda"), None, e(3), e(69), e(10), e(3))
        assert out["body_pose"].shape == (0, 69) and out["loss"].shape == (0,)
        j, v = m.lbs(e(3), e(69), e(10), e(3))
        assert j.shape == (0, 45, 3) and v.shape == (0, 6890, 3)
        return
    j3d, tr0 = _problem(B, seed=5)
    big_j3d, big_tr0 = _problem(300, seed=5)
    out = _fit(j3d, tr0, iters=20)
    ref = _fit(big_j3d, big_tr0, iters=20)                       # same seed: first B frames are identical inputs
    n = min(B, 300)
    for k in ("global_orient", "body_pose", "betas", "transl"):
        assert (out[k][:n] - ref[k][:n]).abs().max() < 2e-5, k


def test_lbs_zero_pose_is_the_shape_blend_at_full_size():
    B = 4096
    c = H.body_consts()
    p = synthetic.make_poses(B, seed=9)
    z = lambda k: torch.zeros(B, k, device="cuda")
    j, v = H.native_model().lbs(z(3), z(69), H.cuda(p.betas), None)
    rows = [0, 1234, B - 1]
    vs = c.v_template[None] + np.einsum("vak,bk->bva", c.shapedirs, p.betas[rows])
    assert np.abs(v[rows].cpu().numpy() - vs).max() < 5e-6
    rest = np.einsum("jv,bva->bja", c.J_regressor, vs)
    assert np.abs(j[rows, :24].cpu().numpy() - rest).max() < 5e-6


def test_lbs_root_rotation_is_rigid_at_full_size():
    B = 1024
    p = synthetic.make_poses(B, seed=13)
    m = H.native_model()
    go, bp, be, tr = map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl))
    j1, v1 = m.lbs(go, bp, be, tr)
    j0, v0 = m.lbs(torch.zeros_like(go), bp, be, None)
    # rotate the unrotated result about its root joint and translate
    ang = go.norm(dim=1, keepdim=True).clamp_min(1e-12)
    u = go / ang
    K = torch.zeros(B, 3, 3, device="cuda")
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -u[:, 2], u[:, 1], u[:, 2], -u[:, 0], -u[:, 1], u[:, 0]
    R = torch.eye(3, device="cuda") + torch.sin(ang)[..., None] * K + (1 - torch.cos(ang))[..., None] * (K @ K)
    root = j0[:, :1]
    moved = torch.einsum("bij,bvj->bvi", R, v0 - root) + root + tr[:, None]
    assert (moved - v1).abs().max() < 2e-5
