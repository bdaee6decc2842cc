"""Motion-file I/O (keypoints2body_amd/io/motion.py) against outputs of the REAL reference's
``io/motion.py`` on the same seeded files (tests/golden/io_motion.npz, oracle/gen_golden_io_eval.py)."""
import io
import warnings
import zipfile

import numpy as np
import pytest
import torch

from keypoints2body_amd.io import load_motion_data, write_smplx_zip, write_smplx_zip_from_smpl_data
from keypoints2body_amd.models.smpl_data import SMPLData
from tests import helpers as H


@pytest.fixture(scope="module")
def g():
    return dict(np.load(H.GOLDEN / "io_motion.npz"))


def _write_inputs(g, tmp_path):
    np.save(tmp_path / "a22.npy", g["in_j22"])
    np.savez(tmp_path / "b24.npz", joints=g["in_j24"], other=np.zeros(3))
    (tmp_path / "c22.csv").write_bytes(g["in_csv_text"].tobytes())


@pytest.mark.parametrize("tag,fname,layout", [("npy22", "a22.npy", None), ("npz24", "b24.npz", None),
                                               ("csv22", "c22.csv", None), ("npy22_explicit", "a22.npy", "AMASS")])
def test_load_motion_data_matches_reference(g, tmp_path, tag, fname, layout):
    _write_inputs(g, tmp_path)
    joints, lay, k = load_motion_data(tmp_path / fname, layout)
    assert lay == str(g[f"{tag}_layout"]) and k == int(g[f"{tag}_k"])
    assert joints.dtype == g[f"{tag}_joints"].dtype
    np.testing.assert_array_equal(joints, g[f"{tag}_joints"])


def test_load_motion_data_errors(tmp_path):
    np.savez(tmp_path / "nojoints.npz", poses=np.zeros((2, 72)))
    with pytest.raises(ValueError, match="Unsupported .npz format"):
        load_motion_data(tmp_path / "nojoints.npz")
    (tmp_path / "x.txt").write_text("1 2 3")
    with pytest.raises(ValueError, match="Unsupported 3D joints file format"):
        load_motion_data(tmp_path / "x.txt")
    np.save(tmp_path / "odd.npy", np.zeros((3, 19, 3)))
    with pytest.raises(ValueError):                      # unknown joint count: the adapter refuses (adapters.py)
        load_motion_data(tmp_path / "odd.npy")


def _read_zip(path):
    out = {}
    with zipfile.ZipFile(path) as zf:
        names = zf.namelist()
        for n in names:
            with np.load(io.BytesIO(zf.read(n))) as d:
                out[n] = {k: d[k] for k in d.keys()}
    return names, out


def test_write_smplx_zip_matches_reference(g, tmp_path):
    zp = write_smplx_zip(tmp_path, g["zip_poses"], g["zip_betas"], g["zip_transl"], zip_name="seq.zip", person_idx=2)
    names, members = _read_zip(zp)
    assert names == [str(n) for n in g["zip_names"]]
    for n in names:
        assert list(members[n].keys()) == [str(k) for k in g["zip_keys"]]
        for k, v in members[n].items():
            ref = g[f"zip::{n}::{k}"]
            assert v.dtype == ref.dtype and v.shape == ref.shape
            np.testing.assert_array_equal(v, ref)


def test_write_smplx_zip_shape_errors(tmp_path):
    with pytest.raises(ValueError, match=r"Expected poses shape \(T,72\)"):
        write_smplx_zip(tmp_path, np.zeros((2, 69)), np.zeros(10), np.zeros((2, 3)))
    with pytest.raises(ValueError, match="Expected betas shape"):
        write_smplx_zip(tmp_path, np.zeros((2, 72)), np.zeros((3, 10)), np.zeros((2, 3)))
    with pytest.raises(ValueError, match="Expected transl shape"):
        write_smplx_zip(tmp_path, np.zeros((2, 72)), np.zeros(10), np.zeros((1, 3)))


def test_write_smplx_zip_from_smpl_data(g, tmp_path):
    poses = torch.tensor(g["zip_poses"], dtype=torch.float32)
    data = SMPLData(betas=torch.tensor(g["zip_betas"]).repeat(3, 1), global_orient=poses[:, :3], body_pose=poses[:, 3:],
                    transl=torch.tensor(g["zip_transl"], dtype=torch.float32))
    zp = write_smplx_zip_from_smpl_data(tmp_path, data, zip_name="d.zip")
    names, members = _read_zip(zp)
    assert names == [f"frame_{t:06d}/person_00.npz" for t in range(3)]
    np.testing.assert_array_equal(members[names[1]]["body_pose"], g["zip_poses"][1, 3:66].astype(np.float32))
    with pytest.raises(ValueError, match="transl is required"):
        write_smplx_zip_from_smpl_data(tmp_path, SMPLData(betas=data.betas, global_orient=data.global_orient,
                                                          body_pose=data.body_pose, transl=None))


def test_layout_conversion_warns(tmp_path):
    """A 25-joint input is reduced to the canonical layout and says so (io/motion.py:51-55)."""
    np.save(tmp_path / "m25.npy", np.random.default_rng(0).normal(size=(2, 25, 3)))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        joints, lay, k = load_motion_data(tmp_path / "m25.npy")
    if k != 25:
        assert any("Converted input joints from 25" in str(x.message) for x in w)
