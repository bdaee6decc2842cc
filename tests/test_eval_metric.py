"""MPJAE metric: CPU restatement vs the real reference's outputs (not gpu), HIP kernel vs both (gpu)."""
import numpy as np
import pytest
import torch

from oracle import eval_np
from tests import helpers as H


@pytest.fixture(scope="module")
def g():
    return dict(np.load(H.GOLDEN / "mpjae.npz"))


def test_oracle_reproduces_reference(g):
    T = g["pred"].shape[0]
    np.testing.assert_array_equal(eval_np.rotvec_to_rotmat(g["gt"].reshape(T, 24, 3)), g["rotmat_gt"])
    ang = eval_np.compute_angular_error_deg(g["pred"].reshape(T, 24, 3), g["gt"].reshape(T, 24, 3))
    np.testing.assert_array_equal(ang, g["angles_deg"])
    mean, total, count = eval_np.evaluate_pose_pair(g["pred"], g["gt"])
    assert (mean, total, count) == (float(g["mean"]), float(g["total"]), int(g["count"]))
    m2, t2, c2 = eval_np.evaluate_pose_pair(np.concatenate([g["pred"], g["pred"][:5]]), g["gt"][:, :66])
    np.testing.assert_allclose([m2, t2, c2], g["ragged"], rtol=0, atol=0)


def test_amass_sequence_loader_and_errors(tmp_path):
    from keypoints2body_amd.evaluation import discover_amass_npz_files, load_amass_sequence, save_prediction_pose
    rng = np.random.default_rng(0)
    (tmp_path / "sub").mkdir()
    np.savez(tmp_path / "sub" / "b.npz", joints=rng.normal(size=(6, 24, 3)), global_orient=rng.normal(size=(5, 3)),
             body_pose=rng.normal(size=(5, 69)))
    np.savez(tmp_path / "a.npz", joints=rng.normal(size=(2, 22, 3)), global_orient=rng.normal(size=3), body_pose=rng.normal(size=69))
    np.savez(tmp_path / "bad.npz", joints=np.zeros((2, 22, 3)))
    files = discover_amass_npz_files(tmp_path)
    assert [f.name for f in files] == ["a.npz", "bad.npz", "b.npz"]
    assert discover_amass_npz_files(tmp_path / "a.npz") == [(tmp_path / "a.npz").resolve()]
    j, p = load_amass_sequence(tmp_path / "sub" / "b.npz")
    assert j.shape == (5, 22, 3) and p.shape == (5, 72) and j.dtype == p.dtype == np.float32
    j, p = load_amass_sequence(tmp_path / "a.npz")          # 1-D pose vectors mean one frame
    assert j.shape == (1, 22, 3) and p.shape == (1, 72)
    with pytest.raises(KeyError, match="Missing keys"):
        load_amass_sequence(tmp_path / "bad.npz")
    save_prediction_pose(p, tmp_path / "sub" / "b.npz", tmp_path, tmp_path / "out")
    assert np.load(tmp_path / "out" / "sub" / "b.npz")["pose"].shape == (1, 72)


def test_metric_needs_a_device_when_none_is_visible():
    if torch.cuda.is_available():
        pytest.skip("a HIP device is visible")
    from keypoints2body_amd.evaluation import evaluate_pose_pair
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP device"):
        evaluate_pose_pair(np.zeros((2, 72), np.float32), np.zeros((2, 72), np.float32))


# Tolerance of the device metric.  arccos amplifies rounding of the trace by 1 / sin(angle): at the clip
# (cos = 1 - 1e-6, angle 0.081 deg) one float32 ulp of the cosine moves the angle by 2.4e-3 deg.  sinf / cosf
# / acosf of the GPU maths library differ from numpy's in the last ulp, so single angles are compared to
# 5e-3 deg + 2e-5 relative, the mean to 2e-4 deg.
ANGLE_ATOL, ANGLE_RTOL, MEAN_ATOL = 5e-3, 2e-5, 2e-4


@pytest.mark.gpu
def test_device_metric_matches_reference(g):
    from keypoints2body_amd.evaluation import compute_angular_error_deg, evaluate_pose_pair
    T = g["pred"].shape[0]
    ang = compute_angular_error_deg(g["pred"].reshape(T, 24, 3), g["gt"].reshape(T, 24, 3)).cpu().numpy()
    np.testing.assert_allclose(ang, g["angles_deg"], rtol=ANGLE_RTOL, atol=ANGLE_ATOL)
    mean, total, count = evaluate_pose_pair(g["pred"], g["gt"])
    assert count == int(g["count"]) and abs(mean - float(g["mean"])) < MEAN_ATOL
    m2, t2, c2 = evaluate_pose_pair(np.concatenate([g["pred"], g["pred"][:5]]), torch.tensor(g["gt"][:, :66]))
    assert c2 == int(g["ragged"][2]) and abs(m2 - g["ragged"][0]) < MEAN_ATOL


@pytest.mark.gpu
def test_device_metric_properties_at_full_size():
    """4096 frames x 24 rotations: zero error against itself up to the clip, symmetry, invariance under a
    common left rotation... checked through the identity angle(R_a, R_b) = |b - a| for rotations about one axis."""
    from keypoints2body_amd.evaluation import compute_angular_error_deg
    from keypoints2body_amd import native
    n = 4096 * 24
    gen = torch.Generator().manual_seed(5)
    axis = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=1)
    a, b = torch.rand(n, 1, generator=gen) * 3.0, torch.rand(n, 1, generator=gen) * 3.0
    e = compute_angular_error_deg(axis * a, axis * b).cpu()
    want = torch.rad2deg((a - b).abs().squeeze(1)).clamp_min(float(np.degrees(np.arccos(1 - 1e-6))))
    assert (e - want).abs().max() < 2e-2
    e_rev = compute_angular_error_deg(axis * b, axis * a).cpu()
    assert torch.equal(e, e_rev)
    assert native.angular_error_deg(torch.zeros(0, 3, device="cuda"), torch.zeros(0, 3, device="cuda")).shape == (0,)
    with pytest.raises(ValueError):
        native.angular_error_deg(torch.zeros(4, 3, device="cuda"), torch.zeros(5, 3, device="cuda"))
