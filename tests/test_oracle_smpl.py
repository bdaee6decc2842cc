"""The oracle's SMPL forward (parity unpinned at the smplx boundary, see its header):
checked against an independent float64 twin, invariants and finite differences."""
import numpy as np
import torch

from keypoints2body_amd import synthetic
from oracle.smpl_torch import TorchSMPL, batch_rodrigues, smpl_forward_np
from tests import helpers as H


def _poses(n, seed=11):
    p = synthetic.make_poses(n, seed=seed)
    t = lambda a: torch.tensor(a)
    return p, dict(global_orient=t(p.global_orient), body_pose=t(p.body_pose), betas=t(p.betas), transl=t(p.transl))


def test_torch_forward_matches_float64_twin():
    p, kw = _poses(3)
    out = H.oracle_model()(**kw)
    assert out.joints.shape == (3, 45, 3) and out.vertices.shape == (3, 6890, 3)
    for f in range(3):
        j, v = smpl_forward_np(H.body_consts(), p.global_orient[f], p.body_pose[f], p.betas[f], p.transl[f])
        assert np.abs(out.joints[f].numpy() - j).max() < 5e-6
        assert np.abs(out.vertices[f].numpy() - v).max() < 5e-6


def test_zero_pose_joints_are_regressed_rest_joints():
    c = H.body_consts()
    betas = torch.tensor(synthetic.make_poses(2, seed=3).betas)
    out = H.oracle_model()(global_orient=torch.zeros(2, 3), body_pose=torch.zeros(2, 69), betas=betas)
    v_shaped = c.v_template[None] + np.einsum("vak,bk->bva", c.shapedirs, betas.numpy())
    rest = np.einsum("jv,bva->bja", c.J_regressor, v_shaped)
    assert np.abs(out.joints[:, :24].numpy() - rest).max() < 2e-6
    assert np.abs(out.vertices.numpy() - v_shaped).max() < 2e-6      # identity pose: no correctives, T = I


def test_root_rotation_is_rigid_and_transl_is_additive():
    _, kw = _poses(2)
    m = H.oracle_model()
    base = m(**kw)
    kw0 = dict(kw, global_orient=torch.zeros(2, 3), transl=None)
    unrot = m(**kw0)
    R = batch_rodrigues(kw["global_orient"])
    root = unrot.joints[:, :1]
    moved = torch.einsum("bij,bkj->bki", R, unrot.vertices - root) + root + kw["transl"][:, None]
    assert (moved - base.vertices).abs().max() < 5e-6
    shifted = m(**dict(kw, transl=kw["transl"] + 0.5))
    assert (shifted.joints - base.joints - 0.5).abs().max() < 1e-6


def test_rodrigues_is_a_rotation_and_handles_zero():
    r = torch.tensor([[0.0, 0.0, 0.0], [0.3, -0.2, 0.9], [3.0, 0.1, -0.2]], dtype=torch.float64)
    R = batch_rodrigues(r)
    eye = torch.eye(3, dtype=torch.float64)
    assert (R @ R.transpose(1, 2) - eye).abs().max() < 1e-7
    assert (R[0] - eye).abs().max() < 1e-12
    assert torch.linalg.det(R).sub(1).abs().max() < 1e-7


def test_joint_gradients_by_finite_differences():
    small = synthetic.make_body_model(seed=2, num_vertices=240)
    m = TorchSMPL(small, dtype=torch.float64)
    p = synthetic.make_poses(1, seed=4)
    go = torch.tensor(p.global_orient, dtype=torch.float64, requires_grad=True)
    bp = torch.tensor(p.body_pose, dtype=torch.float64, requires_grad=True)
    be = torch.tensor(p.betas, dtype=torch.float64, requires_grad=True)
    fn = lambda g, b, s: m(global_orient=g, body_pose=b, betas=s).joints[:, :22].sum(dim=1)
    assert torch.autograd.gradcheck(fn, (go, bp, be), eps=1e-6, atol=1e-6)
