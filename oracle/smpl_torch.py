"""ORACLE (test infrastructure, not product code): CPU SMPL forward.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.

What it restates
----------------
The reference never implements the SMPL forward itself; it calls the
third-party ``smplx`` package (reference ``keypoints2body/api/model_factory.py:5,34-40``,
dependency unpinned in ``pyproject.toml:22``, ``smplx>=0.1.28`` in
``environment.yaml:18``).  ``smplx`` is not installed here and its source is not
under ``/root/reference``, so this file restates the *published* SMPL/LBS
formulation that ``smplx.lbs.lbs`` / ``batch_rodrigues`` / ``batch_rigid_transform``
/ ``SMPL.forward`` implement (Loper et al. 2015):

  1. v_shaped = v_template + shapedirs . beta
  2. J        = J_regressor . v_shaped
  3. R_j      = Rodrigues(theta_j), angle = ||theta_j + 1e-8||
  4. v_posed  = v_shaped + vec(R_1..R_{J-1} - I) . posedirs
  5. G_j      = G_parent(j) . [R_j | J_j - J_parent(j)],  A_j = G_j - [0 | G_j J_j]
  6. v        = (sum_j W_vj A_j) [v_posed; 1]
  7. joints   = cat(G_j[:3,3], v[extra_vertex_ids])
  8. joints, v += transl

PARITY UNPINNED for this forward: no reference test or fixture holds a number
at the ``smplx`` boundary (reference ``tests/test_integration_smoke.py:17-20`` asserts
``params is not None`` only and is skipped without model files).  It is checked
instead by a float64 numpy twin (below), finite differences and invariants in
``tests/test_oracle_smpl.py``.  Everything *downstream* of this forward (loss,
prior, Adam loop) is pinned against the real reference: see ``oracle/gen_golden.py``.

The class duck-types the body-model hook the reference's fitters call
(reference ``core/fitters/world_space.py:174-192,259-278``; attributes read via
``getattr`` in ``core/engine.py:140,146,155,180,194-195``).
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch


def batch_rodrigues(rot_vecs: torch.Tensor) -> torch.Tensor:
    """(N,3) axis-angle -> (N,3,3); angle = ||r + 1e-8|| as in smplx."""
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.cos(angle).unsqueeze(1)
    sin = torch.sin(angle).unsqueeze(1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros_like(rx)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(-1, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device).unsqueeze(0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def rigid_chain(rot_mats: torch.Tensor, joints: torch.Tensor, parents):
    """(B,J,3,3),(B,J,3) -> posed joints (B,J,3), relative transforms (B,J,4,4)."""
    B, J = joints.shape[:2]
    rel = joints.clone()
    rel[:, 1:] = joints[:, 1:] - joints[:, parents[1:]]
    top = torch.cat([rot_mats, rel.unsqueeze(-1)], dim=-1)                    # (B,J,3,4)
    bottom = torch.zeros(B, J, 1, 4, dtype=joints.dtype, device=joints.device)
    bottom[..., 3] = 1
    local = torch.cat([top, bottom], dim=-2)                                   # (B,J,4,4)
    chain = [local[:, 0]]
    for i in range(1, J):
        chain.append(torch.matmul(chain[int(parents[i])], local[:, i]))
    G = torch.stack(chain, dim=1)
    posed = G[:, :, :3, 3]
    j_h = torch.cat([joints, torch.zeros(B, J, 1, dtype=joints.dtype, device=joints.device)], dim=-1)
    shift = torch.matmul(G, j_h.unsqueeze(-1))                                 # (B,J,4,1)
    A = G - torch.nn.functional.pad(shift, [3, 0])
    return posed, A


class TorchSMPL(torch.nn.Module):
    """CPU, autograd-able SMPL-family forward with the smplx call signature."""

    NUM_BODY_JOINTS = 23

    def __init__(self, consts, dtype=torch.float32):
        super().__init__()
        as_t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
        self.register_buffer("v_template", as_t(consts.v_template))
        self.register_buffer("shapedirs", as_t(consts.shapedirs))
        self.register_buffer("posedirs", as_t(consts.posedirs))
        self.register_buffer("J_regressor", as_t(consts.J_regressor))
        self.register_buffer("lbs_weights", as_t(consts.lbs_weights))
        self.register_buffer("parents", torch.as_tensor(np.asarray(consts.parents), dtype=torch.long))
        self.register_buffer(
            "extra_vertex_ids", torch.as_tensor(np.asarray(consts.extra_vertex_ids), dtype=torch.long)
        )
        self.num_betas = int(self.shapedirs.shape[2])
        self.dtype = dtype

    def forward(self, global_orient=None, body_pose=None, betas=None, transl=None,
                return_full_pose=False, return_verts=True, **_unused):
        B = max(x.shape[0] for x in (global_orient, body_pose, betas) if x is not None)
        dev, dt = self.v_template.device, self.dtype
        if global_orient is None:
            global_orient = torch.zeros(B, 3, dtype=dt, device=dev)
        if body_pose is None:
            body_pose = torch.zeros(B, 3 * self.NUM_BODY_JOINTS, dtype=dt, device=dev)
        if betas is None:
            betas = torch.zeros(B, self.num_betas, dtype=dt, device=dev)
        full_pose = torch.cat([global_orient, body_pose], dim=1)
        joints, verts = self._lbs(full_pose, betas, transl)
        return SimpleNamespace(
            vertices=verts, joints=joints, betas=betas, global_orient=global_orient,
            body_pose=body_pose, full_pose=full_pose if return_full_pose else None,
        )

    def _lbs(self, full_pose, shape, transl):
        """Steps 1-8 of the module docstring for a full pose (B, 3J) and all shape coefficients (B, NB)."""
        B = full_pose.shape[0]
        dev, dt = self.v_template.device, self.dtype
        J = self.parents.shape[0]
        v_shaped = self.v_template + torch.einsum("bl,mkl->bmk", shape, self.shapedirs)
        joints_rest = torch.einsum("bik,ji->bjk", v_shaped, self.J_regressor)
        rot = batch_rodrigues(full_pose.reshape(-1, 3)).view(B, J, 3, 3)
        ident = torch.eye(3, dtype=dt, device=dev)
        pose_feature = (rot[:, 1:] - ident).reshape(B, -1)
        v_posed = v_shaped + torch.matmul(pose_feature, self.posedirs).view(B, -1, 3)
        posed_joints, A = rigid_chain(rot, joints_rest, self.parents)
        W = self.lbs_weights.unsqueeze(0).expand(B, -1, -1)
        T = torch.matmul(W, A.reshape(B, J, 16)).view(B, -1, 4, 4)
        ones = torch.ones(B, v_posed.shape[1], 1, dtype=dt, device=dev)
        v_h = torch.matmul(T, torch.cat([v_posed, ones], dim=2).unsqueeze(-1))
        verts = v_h[:, :, :3, 0]
        joints = torch.cat([posed_joints, verts[:, self.extra_vertex_ids]], dim=1)
        if transl is not None:
            joints = joints + transl.unsqueeze(1)
            verts = verts + transl.unsqueeze(1)
        return joints, verts


class TorchSMPLX(TorchSMPL):
    """SMPL-X duck type (smplx ``SMPLX.forward`` with ``use_pca=False``): 55 joints, full pose = [global_orient |
    body_pose 63 | jaw | leye | reye | left_hand_pose 45 | right_hand_pose 45], shape = [betas | expression].
    PARITY UNPINNED at the smplx boundary exactly like ``TorchSMPL``; in addition the reference's own SMPL-X handling
    is inconsistent with smplx (SURVEY.md note N3), so the semantics here are this engine's definition."""

    NUM_BODY_JOINTS = 21
    NUM_HAND_JOINTS = 15

    def __init__(self, consts, dtype=torch.float32, num_betas: int = 10):
        super().__init__(consts, dtype)
        self.num_shape = int(self.shapedirs.shape[2])
        self.num_betas = num_betas
        self.num_expression_coeffs = self.num_shape - num_betas

    def forward(self, global_orient=None, body_pose=None, betas=None, transl=None, expression=None, jaw_pose=None,
                leye_pose=None, reye_pose=None, left_hand_pose=None, right_hand_pose=None,
                return_full_pose=False, return_verts=True, **_unused):
        given = [x for x in (global_orient, body_pose, betas, expression, left_hand_pose) if x is not None]
        B = max(x.shape[0] for x in given)
        dev, dt = self.v_template.device, self.dtype
        z = lambda x, c: torch.zeros(B, c, dtype=dt, device=dev) if x is None else x
        parts = [z(global_orient, 3), z(body_pose, 63), z(jaw_pose, 3), z(leye_pose, 3), z(reye_pose, 3),
                 z(left_hand_pose, 45), z(right_hand_pose, 45)]
        if parts[1].shape[1] != 63:
            raise ValueError(f"SMPL-X body_pose must be (B,63), got {tuple(parts[1].shape)}")
        full_pose = torch.cat(parts, dim=1)
        shape = torch.cat([z(betas, self.num_betas), z(expression, self.num_expression_coeffs)], dim=1)
        joints, verts = self._lbs(full_pose, shape, transl)
        return SimpleNamespace(vertices=verts, joints=joints, betas=shape[:, :self.num_betas], global_orient=parts[0],
                               body_pose=parts[1], full_pose=full_pose if return_full_pose else None)


class TorchSMPLH(TorchSMPL):
    """SMPL-H duck type (smplx ``SMPLH.forward`` with ``use_pca=False``): 52 joints, full pose = [global_orient |
    body_pose 63 | left_hand_pose 45 | right_hand_pose 45], 10 betas.  PARITY UNPINNED at the smplx boundary like ``TorchSMPL``."""

    NUM_BODY_JOINTS = 21
    NUM_HAND_JOINTS = 15

    def __init__(self, consts, dtype=torch.float32):
        super().__init__(consts, dtype)
        self.num_betas = int(self.shapedirs.shape[2])

    def forward(self, global_orient=None, body_pose=None, betas=None, transl=None, left_hand_pose=None, right_hand_pose=None,
                return_full_pose=False, return_verts=True, **_unused):
        given = [x for x in (global_orient, body_pose, betas, left_hand_pose) if x is not None]
        B = max(x.shape[0] for x in given)
        dev, dt = self.v_template.device, self.dtype
        z = lambda x, c: torch.zeros(B, c, dtype=dt, device=dev) if x is None else x
        parts = [z(global_orient, 3), z(body_pose, 63), z(left_hand_pose, 45), z(right_hand_pose, 45)]
        if parts[1].shape[1] != 63:
            raise ValueError(f"SMPL-H body_pose must be (B,63), got {tuple(parts[1].shape)}")
        full_pose = torch.cat(parts, dim=1)
        shape = z(betas, self.num_betas)
        joints, verts = self._lbs(full_pose, shape, transl)
        return SimpleNamespace(vertices=verts, joints=joints, betas=shape, global_orient=parts[0], body_pose=parts[1],
                               full_pose=full_pose if return_full_pose else None)


# --------------------------------------------------------------------------
# float64 numpy twin: an independent, loop-style restatement used to check the
# torch module above (different code path: explicit per-joint loops).
# --------------------------------------------------------------------------
def rodrigues_np(r: np.ndarray) -> np.ndarray:
    r = np.asarray(r, dtype=np.float64)
    angle = np.sqrt(((r + 1e-8) ** 2).sum())
    u = r / angle
    K = np.array([[0.0, -u[2], u[1]], [u[2], 0.0, -u[0]], [-u[1], u[0], 0.0]])
    return np.eye(3) + np.sin(angle) * K + (1.0 - np.cos(angle)) * (K @ K)


def smpl_forward_np(consts, global_orient, body_pose, betas, transl=None):
    """Single-frame float64 forward; returns (joints (J+E,3), vertices (V,3))."""
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    v_t, S, Pd = f64(consts.v_template), f64(consts.shapedirs), f64(consts.posedirs)
    Jr, W = f64(consts.J_regressor), f64(consts.lbs_weights)
    parents = np.asarray(consts.parents)
    J = parents.shape[0]
    theta = np.concatenate([f64(global_orient).ravel(), f64(body_pose).ravel()]).reshape(J, 3)
    beta = f64(betas).ravel()

    v_shaped = v_t + S @ beta
    j_rest = Jr @ v_shaped
    R = np.stack([rodrigues_np(theta[j]) for j in range(J)])
    feat = (R[1:] - np.eye(3)).reshape(-1)
    v_posed = v_shaped + (feat @ Pd).reshape(-1, 3)

    Rg = np.zeros((J, 3, 3))
    pg = np.zeros((J, 3))
    Rg[0], pg[0] = R[0], j_rest[0]
    for j in range(1, J):
        p = parents[j]
        Rg[j] = Rg[p] @ R[j]
        pg[j] = pg[p] + Rg[p] @ (j_rest[j] - j_rest[p])
    A_rot = Rg
    A_tr = pg - np.einsum("jab,jb->ja", Rg, j_rest)
    verts = np.zeros_like(v_posed)
    for j in range(J):
        verts += W[:, j:j + 1] * (v_posed @ A_rot[j].T + A_tr[j])
    joints = np.concatenate([pg, verts[np.asarray(consts.extra_vertex_ids)]], axis=0)
    if transl is not None:
        t = f64(transl).ravel()
        joints = joints + t
        verts = verts + t
    return joints, verts
