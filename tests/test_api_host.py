"""API-shape tests mirrored from the reference's own test-suite (reference
``tests/test_adapters.py``, ``tests/test_models.py``, ``tests/test_api_surface.py``): same
shapes, labels and container behaviour, against this package.  CPU only."""
import numpy as np
import pytest
import torch

from keypoints2body_amd import (FLAMEData, MANOData, SMPLData, SMPLXData, optimize_params_frame,
                                optimize_params_sequence, optimize_shape_sequence)
from keypoints2body_amd.core.joints.adapters import (ADAPTERS, adapt_layout, adapt_layout_and_conf,
                                                     normalize_frame_observations, normalize_joints_frame,
                                                     normalize_joints_sequence, normalize_sequence_observations,
                                                     resolve_adapter)


def test_api_exports_exist():
    assert callable(optimize_params_frame) and callable(optimize_params_sequence) and callable(optimize_shape_sequence)


def test_normalize_joints_frame_k3_and_k4():
    j3d, conf = normalize_joints_frame(np.zeros((22, 3), dtype=np.float32))
    assert tuple(j3d.shape) == (1, 22, 3) and tuple(conf.shape) == (22,) and torch.all(conf == 1)
    xyzc = np.zeros((22, 4), dtype=np.float32)
    xyzc[:, 3] = 0.7
    j3d, conf = normalize_joints_frame(xyzc)
    assert tuple(j3d.shape) == (1, 22, 3) and torch.allclose(conf, torch.full((22,), 0.7))
    with pytest.raises(ValueError):
        normalize_joints_frame(np.zeros((22, 5), dtype=np.float32))
    with pytest.raises(ValueError):
        normalize_joints_frame([[0, 0, 0]])


def test_normalize_joints_sequence_tk4():
    seq = np.zeros((5, 22, 4), dtype=np.float32)
    seq[:, :, 3] = 0.2
    xyz, conf = normalize_joints_sequence(seq)
    assert tuple(xyz.shape) == (5, 22, 3) and tuple(conf.shape) == (5, 22) and torch.allclose(conf, torch.tensor(0.2))


def test_adapt_layout_manny_to_amass_and_errors():
    out, layout = adapt_layout(np.zeros((3, 25, 3), dtype=np.float32), "Manny25")
    assert tuple(out.shape) == (3, 22, 3) and layout == "AMASS"
    assert resolve_adapter(22, None).name == "AMASS" and resolve_adapter(24, None).name == "SMPL24"
    with pytest.raises(ValueError):
        resolve_adapter(23, None)
    with pytest.raises(ValueError):
        resolve_adapter(22, "Manny25")
    with pytest.raises(ValueError):
        resolve_adapter(22, "NoSuchLayout")
    assert set(ADAPTERS) == {"SMPL24", "AMASS", "Manny25", "Halpe26", "SpineTrack37"}


def test_adapter_mapping_rotation_and_conf_follow_the_same_indexing():
    rng = np.random.default_rng(0)
    seq = rng.normal(size=(2, 25, 3)).astype(np.float32)
    conf = rng.uniform(size=(2, 25)).astype(np.float32)
    pts, cf, layout = adapt_layout_and_conf(seq, conf, "Manny25")
    src = list(ADAPTERS["Manny25"].mapping)
    assert layout == "AMASS" and np.array_equal(cf, conf[:, src])
    # OpenSim (x, y, z) -> SMPL (z, y, -x)
    assert np.allclose(pts[..., 0], seq[:, src, 2]) and np.allclose(pts[..., 1], seq[:, src, 1])
    assert np.allclose(pts[..., 2], -seq[:, src, 0])
    same, _, lay = adapt_layout_and_conf(seq[:, :22], conf[:, :22], "AMASS")
    assert lay == "AMASS" and np.array_equal(same, seq[:, :22])


def test_normalize_frame_observations_dict_smplx():
    obs = {"body": np.zeros((22, 4), dtype=np.float32), "left_hand": np.zeros((21, 3), dtype=np.float32),
           "right_hand": np.zeros((21, 3), dtype=np.float32), "face": np.zeros((10, 3), dtype=np.float32)}
    j3d, conf, model_idx, out_layout = normalize_frame_observations(obs, layout=None, body_model="smplx")
    assert tuple(j3d.shape) == (1, 74, 3) and tuple(conf.shape) == (74,) and tuple(model_idx.shape) == (74,)
    assert out_layout == "GENERIC"
    assert model_idx[:22].tolist() == list(range(22)) and model_idx[22].item() == 25 and model_idx[43].item() == 46
    assert model_idx[64].item() == 67
    with pytest.raises(ValueError):
        normalize_frame_observations({"left_hand": np.zeros((21, 3), np.float32)}, layout=None, body_model="smpl")
    with pytest.raises(ValueError):
        normalize_frame_observations({}, layout=None, body_model="smplx")


def test_normalize_sequence_observations_dict_smplh():
    obs = {"body": np.zeros((4, 22, 3), dtype=np.float32), "left_hand": np.zeros((4, 21, 3), dtype=np.float32),
           "right_hand": np.zeros((4, 21, 3), dtype=np.float32)}
    xyz, conf, model_idx, out_layout = normalize_sequence_observations(obs, layout=None, body_model="smplh")
    assert tuple(xyz.shape) == (4, 64, 3) and tuple(conf.shape) == (4, 64) and tuple(model_idx.shape) == (64,)
    assert out_layout == "GENERIC"
    with pytest.raises(ValueError):
        normalize_sequence_observations({"body": np.zeros((4, 22, 3), np.float32),
                                         "left_hand": np.zeros((3, 21, 3), np.float32)}, layout=None,
                                        body_model="smplh")


def test_pose_concat_validate_detach_numpy():
    p = SMPLData(betas=torch.zeros(1, 10), global_orient=torch.zeros(1, 3), body_pose=torch.zeros(1, 69),
                 transl=torch.zeros(1, 3))
    p.validate()
    assert tuple(p.pose.shape) == (1, 72)
    g = SMPLData(betas=torch.zeros(1, 10, requires_grad=True), global_orient=torch.zeros(1, 3, requires_grad=True),
                 body_pose=torch.zeros(1, 69, requires_grad=True))
    d = g.detach()
    assert isinstance(d.betas, torch.Tensor) and d.betas.requires_grad is False and d.transl is None
    n = SMPLData(betas=np.zeros((1, 10), np.float32), global_orient=np.zeros((1, 3), np.float32),
                 body_pose=np.zeros((1, 69), np.float32))
    assert n.pose.shape == (1, 72) and n.to(torch.device("cpu")).betas is n.betas
    with pytest.raises(ValueError):
        SMPLData(betas=None, global_orient=torch.zeros(1, 3), body_pose=torch.zeros(1, 69)).validate()
    x = SMPLXData(betas=torch.zeros(1, 10), global_orient=torch.zeros(1, 3), body_pose=torch.zeros(1, 69),
                  jaw_pose=torch.zeros(1, 3), metadata={"k": 1})
    assert x.detach().metadata == {"k": 1} and x.to(torch.device("cpu")).jaw_pose is x.jaw_pose


def test_mano_and_flame_data_containers():
    mano = MANOData(betas=torch.zeros(1, 10), global_orient=torch.zeros(1, 3), body_pose=torch.zeros(1, 0),
                    hand_pose=torch.zeros(1, 45), transl=torch.zeros(1, 3))
    flame = FLAMEData(betas=torch.zeros(1, 10), global_orient=torch.zeros(1, 3), body_pose=torch.zeros(1, 0),
                      expression=torch.zeros(1, 10), jaw_pose=torch.zeros(1, 3), transl=torch.zeros(1, 3))
    mano.validate()
    flame.validate()
    assert tuple(mano.pose.shape) == (1, 3) and tuple(flame.pose.shape) == (1, 3)


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_api_error_conventions_without_device():
    j = np.zeros((22, 3), np.float32)
    with pytest.raises(RuntimeError, match="no CPU"):
        optimize_params_frame(j, joint_layout="AMASS")
    with pytest.raises(RuntimeError):
        optimize_params_frame(j, joint_layout="AMASS", device="cpu")
