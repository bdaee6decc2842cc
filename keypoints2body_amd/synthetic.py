"""Deterministic synthetic assets for the fitting path.

The licensed SMPL model, the ``gmm_08.pkl`` pose prior and the mean-parameter
file the reference expects (reference ``docs/getting_started.rst:36-57``) are
not redistributable, so every test, benchmark and golden fixture in this
repository runs on *synthetic* stand-ins of the same shapes:

* an SMPL-shaped body model: V=6890 vertices, 24 joints on the real SMPL
  kinematic tree, 10 shape directions, 207 pose-corrective directions,
  45 output joints (24 + 21 vertex-selected);
* an 8-component, 69-D Gaussian-mixture pose prior;
* seeded target joints for a sequence of frames.

Everything is produced by a counter-based integer hash followed by IEEE
add/mul/div only (no libm calls), so this container and the GPU box generate
bit-identical arrays.  ``checksum`` gives a cheap fingerprint that the golden
fixtures record.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

# Real SMPL kinematic tree; joint 0 is the root.
SMPL_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
    dtype=np.int32,
)

# Hand-written T-pose skeleton (metres, y up, +x = subject's left).  Only has
# to be a plausible human tree; it is not the licensed SMPL template.
_REST_JOINTS = np.array(
    [
        [0.000, -0.230, 0.020],
        [0.070, -0.320, 0.010],
        [-0.070, -0.320, 0.010],
        [0.000, -0.110, -0.010],
        [0.100, -0.700, 0.020],
        [-0.100, -0.700, 0.020],
        [0.000, 0.020, 0.010],
        [0.090, -1.100, -0.020],
        [-0.090, -1.100, -0.020],
        [0.000, 0.080, 0.020],
        [0.110, -1.160, 0.100],
        [-0.110, -1.160, 0.100],
        [0.000, 0.290, -0.020],
        [0.080, 0.200, -0.010],
        [-0.080, 0.200, -0.010],
        [0.000, 0.380, 0.030],
        [0.180, 0.230, -0.020],
        [-0.180, 0.230, -0.020],
        [0.440, 0.220, -0.030],
        [-0.440, 0.220, -0.030],
        [0.690, 0.220, -0.010],
        [-0.690, 0.220, -0.010],
        [0.780, 0.210, -0.010],
        [-0.780, 0.210, -0.010],
    ],
    dtype=np.float64,
)

# SMPL-X kinematic tree (55 joints, smplx numbering): 0-21 body, 22 jaw, 23/24 eyes (children of the head, 15),
# 25-39 left hand (index, middle, pinky, ring, thumb: three joints each, rooted at the left wrist, 20), 40-54 right hand.
SMPLX_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
     20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
     21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53],
    dtype=np.int32,
)


# SMPL-H kinematic tree (52 joints, smplx numbering): the SMPL-X tree without jaw and eyes - 0-21 body, 22-36 left hand,
# 37-51 right hand.
SMPLH_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
     20, 22, 23, 20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35,
     21, 37, 38, 21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50],
    dtype=np.int32,
)


def _smplh_rest_joints() -> np.ndarray:
    """The 55-joint T-pose skeleton below without its jaw and eye joints."""
    r = _smplx_rest_joints()
    return np.concatenate([r[:22], r[25:]], axis=0)


def _smplx_rest_joints() -> np.ndarray:
    """Plausible T-pose skeleton of the 55-joint tree: the SMPL body joints 0-21, jaw and eyes on the head, five
    three-joint fingers fanning out of either wrist (hand-written, not a licensed template)."""
    rest = np.zeros((55, 3), dtype=np.float64)
    rest[:22] = _REST_JOINTS[:22]
    head = rest[15]
    rest[22] = head + np.array([0.0, -0.02, 0.05])       # jaw
    rest[23] = head + np.array([0.03, 0.06, 0.08])       # left eye
    rest[24] = head + np.array([-0.03, 0.06, 0.08])      # right eye
    for side, wrist, first in ((1.0, 20, 25), (-1.0, 21, 40)):
        for fi, (dy, dz) in enumerate(((0.012, 0.025), (0.014, 0.005), (0.004, -0.030), (0.010, -0.012), (-0.010, 0.035))):
            base = rest[wrist] + np.array([side * 0.085, dy, dz])
            for seg in range(3):
                rest[first + 3 * fi + seg] = base + np.array([side * 0.028 * seg, -0.002 * seg, 0.0])
    return rest


_SQRT3 = 1.7320508075688772  # literal, so no libm sqrt is involved
_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(stream: int, n: int, seed: int = 0) -> np.ndarray:
    """n reproducible float64 draws in [0, 1) from (seed, stream, counter)."""
    with np.errstate(over="ignore"):
        base = _mix64(
            np.array([np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream)], dtype=np.uint64)
        )[0]
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base
    bits = _mix64(ctr) >> np.uint64(11)
    return bits.astype(np.float64) * (1.0 / 9007199254740992.0)


def normalish(stream: int, shape, seed: int = 0) -> np.ndarray:
    """Zero-mean, unit-variance, bell-shaped draws (centred sum of 4 uniforms)."""
    n = int(np.prod(shape))
    u = uniform(stream, 4 * n, seed).reshape(n, 4)
    s = ((u[:, 0] + u[:, 1]) + (u[:, 2] + u[:, 3])) - 2.0
    return (s * _SQRT3).reshape(shape)


def _row_sums(a: np.ndarray) -> np.ndarray:
    """Strictly left-to-right row sums (cumsum), independent of SIMD width."""
    return np.cumsum(a, axis=1)[:, -1:]


def checksum(*arrays: np.ndarray) -> int:
    """Order-sensitive 64-bit fingerprint of the raw bytes of the arrays."""
    acc = np.uint64(0x243F6A8885A308D3)
    for a in arrays:
        raw = np.ascontiguousarray(a).view(np.uint8)
        pad = (-raw.size) % 8
        if pad:
            raw = np.concatenate([raw.ravel(), np.zeros(pad, np.uint8)])
        words = raw.ravel().view(np.uint64)
        idx = np.arange(words.size, dtype=np.uint64)
        with np.errstate(over="ignore"):
            h = _mix64(words ^ (idx * np.uint64(0x9E3779B97F4A7C15)))
            acc = _mix64(np.array([acc ^ np.bitwise_xor.reduce(h) ^ np.uint64(words.size)], dtype=np.uint64))[0]
    return int(acc)


@dataclass
class SyntheticBodyModel:
    """Constant tensors of an SMPL-shaped model, in the layouts ``smplx`` exposes.

    Shapes follow the reference's body-model hook:
    ``v_template (V,3)``, ``shapedirs (V,3,NB)``, ``posedirs (9*(J-1), 3V)``,
    ``J_regressor (J,V)``, ``lbs_weights (V,J)``, ``parents (J,)``,
    ``extra_vertex_ids (E,)`` (joints J..J+E-1 are these vertices).
    """

    v_template: np.ndarray
    shapedirs: np.ndarray
    posedirs: np.ndarray
    J_regressor: np.ndarray
    lbs_weights: np.ndarray
    parents: np.ndarray
    extra_vertex_ids: np.ndarray
    seed: int = 0

    @property
    def num_vertices(self) -> int:
        return int(self.v_template.shape[0])

    @property
    def num_joints(self) -> int:
        return int(self.parents.shape[0])

    @property
    def num_betas(self) -> int:
        return int(self.shapedirs.shape[2])

    def fingerprint(self) -> int:
        return checksum(
            self.v_template, self.shapedirs, self.posedirs, self.J_regressor,
            self.lbs_weights, self.parents, self.extra_vertex_ids,
        )


def make_body_model_x(seed: int = 0, num_vertices: int = 10475, num_shape: int = 20, num_extra: int = 72) -> SyntheticBodyModel:
    """SMPL-X-shaped synthetic model: V = 10475, 55 joints on the SMPL-X tree, 20 shape coefficients (10 betas | 10
    expression coefficients, concatenated as smplx does), 486 pose-corrective rows, 55 + 72 = 127 output joints."""
    return make_body_model(seed, num_vertices, num_shape, parents=SMPLX_PARENTS.copy(), rest=_smplx_rest_joints(),
                           num_extra=num_extra)


def make_body_model_h(seed: int = 0, num_vertices: int = 6890, num_betas: int = 10, num_extra: int = 21) -> SyntheticBodyModel:
    """SMPL-H-shaped synthetic model: V = 6890, 52 joints (body + two 15-joint hands), 10 betas, 459 pose-corrective rows."""
    return make_body_model(seed, num_vertices, num_betas, parents=SMPLH_PARENTS.copy(), rest=_smplh_rest_joints(),
                           num_extra=num_extra)


def make_body_model(seed: int = 0, num_vertices: int = 6890, num_betas: int = 10, parents=None, rest=None,
                    num_extra: int = 21) -> SyntheticBodyModel:
    """Build the SMPL-shaped synthetic model (float32 arrays); `parents` / `rest` select another tree."""
    parents = SMPL_PARENTS.copy() if parents is None else parents
    J = parents.shape[0]
    V = num_vertices
    rest = _REST_JOINTS if rest is None else rest

    vid = np.arange(V)
    primary = vid % J                      # joint each vertex hangs on
    ring = ((vid // J) % 2) == 0           # "ring" vertices sit around the joint
    par_of = np.where(parents[primary] < 0, primary, parents[primary])

    u_bone = uniform(1, V, seed)
    noise = normalish(2, (V, 3), seed)
    along = np.where(ring, 0.0, u_bone)[:, None]
    radius = np.where(ring, 0.05, 0.04)[:, None]
    v_template = rest[primary] + along * (rest[par_of] - rest[primary]) + radius * noise

    # Joint regressor: each joint is a convex combination of its ring vertices.
    u_reg = uniform(3, V, seed)
    w_reg = np.where(ring, (0.25 + u_reg) * (0.25 + u_reg), 0.0)
    J_regressor = np.zeros((J, V), dtype=np.float64)
    J_regressor[primary, vid] = w_reg
    J_regressor /= _row_sums(J_regressor)

    # Skinning weights: <= 4 non-zeros per vertex like real SMPL, stored dense.
    u_w = uniform(4, 3 * V, seed).reshape(V, 3)
    other_a = (primary + 1 + (uniform(5, V, seed) * (J - 1)).astype(np.int64)) % J
    other_b = (primary + 1 + (uniform(6, V, seed) * (J - 1)).astype(np.int64)) % J
    w_primary = np.where(ring, 0.7 + 0.3 * u_w[:, 0], 1.0 - 0.8 * u_bone)
    w_parent = 1.0 - w_primary
    lbs_weights = np.zeros((V, J), dtype=np.float64)
    np.add.at(lbs_weights, (vid, primary), w_primary)
    np.add.at(lbs_weights, (vid, par_of), w_parent)
    np.add.at(lbs_weights, (vid, other_a), 0.05 * u_w[:, 1])
    np.add.at(lbs_weights, (vid, other_b), 0.03 * u_w[:, 2])
    lbs_weights /= _row_sums(lbs_weights)

    # Shape directions: a few coherent body-scale modes plus per-vertex noise,
    # so that betas are observable from joints.
    shapedirs = 0.004 * normalish(7, (V, 3, num_betas), seed)
    axis_scale = np.array(
        [[1.0, 1.0, 1.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0],
         [1.0, -1.0, 0.0], [0.5, 0.5, -1.0], [1.0, 0.0, -1.0], [0.0, 1.0, 1.0],
         [-0.5, 1.0, 0.5], [1.0, 0.5, 0.5]]
    )
    amp = np.array([0.030, 0.040, 0.030, 0.020, 0.015, 0.012, 0.010, 0.010, 0.008, 0.008])
    for k in range(num_betas):
        shapedirs[:, :, k] += amp[k % 10] * axis_scale[k % 10][None, :] * v_template

    P = 9 * (J - 1)
    posedirs = 0.002 * normalish(8, (P, 3 * V), seed)

    extra_vertex_ids = ((331 + 311 * np.arange(num_extra)) % V).astype(np.int32)

    return SyntheticBodyModel(
        v_template=v_template.astype(np.float32),
        shapedirs=shapedirs.astype(np.float32),
        posedirs=posedirs.astype(np.float32),
        J_regressor=J_regressor.astype(np.float32),
        lbs_weights=lbs_weights.astype(np.float32),
        parents=parents,
        extra_vertex_ids=extra_vertex_ids,
        seed=seed,
    )


@dataclass
class SyntheticGMM:
    """Mixture parameters in the dict layout of ``gmm_08.pkl``
    (reference ``core/prior.py:136-139``): float64 means/covars/weights."""

    means: np.ndarray    # (M, D)
    covars: np.ndarray   # (M, D, D)
    weights: np.ndarray  # (M,)


def make_gmm(seed: int = 0, num_gaussians: int = 8, dim: int = 69) -> SyntheticGMM:
    means = 0.15 * normalish(20, (num_gaussians, dim), seed)
    A = 0.08 * normalish(21, (num_gaussians, dim, 24), seed)  # low rank + diagonal: cond ~ 1e2-1e3
    diag = 0.003 + 0.05 * uniform(22, num_gaussians * dim, seed).reshape(num_gaussians, dim)
    covars = np.einsum("mik,mjk->mij", A, A)
    covars = 0.5 * (covars + np.transpose(covars, (0, 2, 1)))
    covars[:, np.arange(dim), np.arange(dim)] += diag
    w = 1.0 + uniform(23, num_gaussians, seed)
    return SyntheticGMM(means=means, covars=covars, weights=w / _row_sums(w[None, :])[0, 0])


@dataclass
class SyntheticPoses:
    """Ground-truth parameters used to pose the model into target joints."""

    global_orient: np.ndarray  # (T,3)
    body_pose: np.ndarray      # (T,69)
    betas: np.ndarray          # (T,10)
    transl: np.ndarray         # (T,3)


def make_poses(num_frames: int, seed: int = 0, num_betas: int = 10) -> SyntheticPoses:
    """theta ~ 0.2 (root 0.3), beta ~ 0.5, transl ~ 1.0."""
    T = num_frames
    return SyntheticPoses(
        global_orient=(0.3 * normalish(30, (T, 3), seed)).astype(np.float32),
        body_pose=(0.2 * normalish(31, (T, 69), seed)).astype(np.float32),
        betas=(0.5 * normalish(32, (T, num_betas), seed)).astype(np.float32),
        transl=(1.0 * normalish(33, (T, 3), seed)).astype(np.float32),
    )


@dataclass
class SyntheticPosesX:
    """Ground-truth SMPL-X parameters (smplx field names; hands as full axis-angle poses, use_pca=False)."""

    global_orient: np.ndarray    # (T,3)
    body_pose: np.ndarray        # (T,63)
    jaw_pose: np.ndarray         # (T,3)
    leye_pose: np.ndarray        # (T,3)
    reye_pose: np.ndarray        # (T,3)
    left_hand_pose: np.ndarray   # (T,45)
    right_hand_pose: np.ndarray  # (T,45)
    betas: np.ndarray            # (T,10)
    expression: np.ndarray       # (T,10)
    transl: np.ndarray           # (T,3)


def make_poses_x(num_frames: int, seed: int = 0) -> SyntheticPosesX:
    T = num_frames
    f = lambda stream, cols, scale: (scale * normalish(stream, (T, cols), seed)).astype(np.float32)
    return SyntheticPosesX(
        global_orient=f(40, 3, 0.3), body_pose=f(41, 63, 0.2), jaw_pose=f(42, 3, 0.1), leye_pose=f(43, 3, 0.05),
        reye_pose=f(44, 3, 0.05), left_hand_pose=f(45, 45, 0.15), right_hand_pose=f(46, 45, 0.15),
        betas=f(47, 10, 0.5), expression=f(48, 10, 0.5), transl=f(49, 3, 1.0))


@dataclass
class SyntheticPosesH:
    """Ground-truth SMPL-H parameters (smplx field names; hands as full axis-angle poses, use_pca=False)."""

    global_orient: np.ndarray    # (T,3)
    body_pose: np.ndarray        # (T,63)
    left_hand_pose: np.ndarray   # (T,45)
    right_hand_pose: np.ndarray  # (T,45)
    betas: np.ndarray            # (T,10)
    transl: np.ndarray           # (T,3)


def make_poses_h(num_frames: int, seed: int = 0) -> SyntheticPosesH:
    T = num_frames
    f = lambda stream, cols, scale: (scale * normalish(stream, (T, cols), seed)).astype(np.float32)
    return SyntheticPosesH(global_orient=f(50, 3, 0.3), body_pose=f(51, 63, 0.2), left_hand_pose=f(52, 45, 0.15),
                           right_hand_pose=f(53, 45, 0.15), betas=f(54, 10, 0.5), transl=f(55, 3, 1.0))


def target_noise(num_frames: int, num_joints: int, seed: int = 0, scale: float = 0.005) -> np.ndarray:
    """Optional observation noise added to target joints (metres)."""
    return (scale * normalish(34, (num_frames, num_joints, 3), seed)).astype(np.float32)
