"""Host-side logic that runs without a GPU: prior preparation, C-ABI surface, config
conversion, loud failure when there is no device."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from keypoints2body_amd import native
from keypoints2body_amd.core import config as cfgmod
from keypoints2body_amd.prior import MixtureBuffers
from tests import helpers as H

REPO = Path(__file__).resolve().parents[1]


def test_mixture_buffers_match_reference_derivation():
    g = H.gmm_fixture()
    b = MixtureBuffers.from_mixture(g["means"], g["covars"].astype(np.float64), g["weights"])
    assert np.array_equal(b.means, g["ref_means"])
    assert np.abs(b.precisions - g["ref_precisions"]).max() <= 1e-4 * np.abs(g["ref_precisions"]).max()
    np.testing.assert_allclose(b.nll_weights, g["ref_nll_weights"].reshape(-1), rtol=1e-5)
    with pytest.raises(ValueError):
        MixtureBuffers.from_mixture(g["means"], g["covars"][:, :5], g["weights"])


def test_missing_prior_file_raises_instead_of_exiting():
    # the reference prints and calls sys.exit(-1) (core/prior.py:126-131); a library raises instead
    with pytest.raises(FileNotFoundError):
        MixtureBuffers.from_file("/nonexistent/gmm_08.pkl")


def test_prior_pickle_loads_arrays_and_refuses_code(tmp_path):
    """The reference's gmm_XX.pkl is a Python-2 pickle of a dict of numpy arrays (prior.py:133-139): it must load,
    bit-identically to the .npz form, through an unpickler that resolves numpy array globals only; a pickle that
    names any other global (an sklearn object, os.system, ...) must be refused without being executed."""
    import pickle
    g = H.gmm_fixture()
    covars = g["covars"].astype(np.float64)
    good = tmp_path / "gmm_08.pkl"
    with open(good, "wb") as f:
        pickle.dump({"means": g["means"], "covars": covars, "weights": g["weights"]}, f, protocol=2)
    b = MixtureBuffers.from_file(str(good))
    np.savez(tmp_path / "gmm.npz", means=g["means"], covars=covars, weights=g["weights"])
    z = MixtureBuffers.from_file(str(tmp_path / "gmm.npz"))
    assert np.array_equal(b.means, z.means) and np.array_equal(b.precisions, z.precisions)
    assert np.array_equal(b.nll_weights, z.nll_weights) and np.isfinite(b.nll_weights).all()

    marker = tmp_path / "executed"

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, (f"touch {marker}",))

    bad = tmp_path / "evil.pkl"
    with open(bad, "wb") as f:
        pickle.dump({"means": Evil()}, f, protocol=2)
    with pytest.raises(ValueError, match="only plain dicts of numpy arrays"):
        MixtureBuffers.from_file(str(bad))
    assert not marker.exists()
    with open(bad, "wb") as f:
        pickle.dump([1, 2, 3], f, protocol=2)
    with pytest.raises(ValueError, match="Unknown content"):
        MixtureBuffers.from_file(str(bad))


def test_c_abi_exports_every_declared_symbol():
    header = (REPO / "include" / "k2b.h").read_text()
    declared = set(re.findall(r"\b(k2b_[a-z_]+)\s*\(", header))
    assert declared == set(native.EXPORTED_SYMBOLS)
    lib = native.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.k2b_version() >> 16 == 1


def test_header_is_plain_c_and_struct_size_matches(tmp_path):
    """include/k2b.h is the drop-in boundary: it must compile as C99 and as C++ with nothing but the
    standard headers, and sizeof(k2b_fit_config) seen by a C compiler must equal what the library and the
    ctypes mirror use."""
    import shutil, subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text('#include <stdio.h>\n#include "k2b.h"\nint main(void) { printf("%u\\n", (unsigned)sizeof(k2b_fit_config)); return 0; }\n')
    inc = str(native.library_path().parents[2] / "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, str(src), "-o", str(tmp_path / "hdr")], check=True)
    subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)], check=True)
    size = int(subprocess.run([str(tmp_path / "hdr")], check=True, capture_output=True, text=True).stdout)
    import ctypes
    assert size == ctypes.sizeof(native.FitConfigC) == native.load_library().k2b_fit_config_size()


def test_default_fit_config_carries_reference_weights():
    c = native.default_fit_config()
    assert (c.num_iters, c.step_size, c.adam_beta1, c.adam_beta2, c.adam_eps) == (30, 1e-2, 0.9, 0.999, 1e-8)
    assert c.sigma == 100.0 and c.joint_loss_weight == 600.0 and c.shape_prior_weight == 5.0
    assert abs(c.pose_prior_weight - 4.78 * 1.5) < 1e-6 and abs(c.angle_prior_weight - 15.2) < 1e-6
    assert list(c.angle_prior_index) == [52, 55, 9, 12] and list(c.angle_prior_sign) == [1.0, -1.0, -1.0, -1.0]
    assert ctypes.sizeof(native.FitConfigC) == native.load_library().k2b_fit_config_size()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-device failure mode")
def test_no_device_fails_loudly_not_silently():
    c = H.body_consts()
    with pytest.raises(RuntimeError, match="no CPU"):
        native.NativeModel(c.v_template, c.shapedirs, c.posedirs, c.J_regressor, c.lbs_weights, c.parents,
                           c.extra_vertex_ids)
    # straight through the C ABI: k2b_model_create must report K2B_ERR_NO_DEVICE, not compute on the host
    lib = native.load_library()
    h = ctypes.c_void_p()
    f32 = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(ctypes.c_void_p)
    i32 = lambda a: np.ascontiguousarray(a, np.int32).ctypes.data_as(ctypes.c_void_p)
    rc = lib.k2b_model_create(ctypes.byref(h), 6890, 24, 10, 21, f32(c.v_template), f32(c.shapedirs), f32(c.posedirs),
                              f32(c.J_regressor), f32(c.lbs_weights), i32(c.parents), i32(c.extra_vertex_ids))
    assert rc == native.K2B_ERR_NO_DEVICE and b"no HIP device" in lib.k2b_last_error()
    bad = np.array(c.parents, np.int32); bad[5] = 9
    rc = lib.k2b_model_create(ctypes.byref(h), 6890, 24, 10, 21, f32(c.v_template), f32(c.shapedirs), f32(c.posedirs),
                              f32(c.J_regressor), f32(c.lbs_weights), i32(bad), i32(c.extra_vertex_ids))
    assert rc == native.K2B_ERR_INVALID_ARGUMENT


def test_config_conversion_matches_reference_rules():
    f = cfgmod.frame_config_from({"use_lbfgs": False, "num_iters_first": 100})
    assert f.use_lbfgs is False and f.num_iters_first == 100 and f.joint_loss_weight == 600.0
    s = cfgmod.sequence_config_from({"frame": {"use_lbfgs": False}, "use_shape_optimization": False, "fix_foot": True})
    assert s.frame.use_lbfgs is False and s.use_shape_optimization is False and s.fix_foot and s.num_shape_frames == 50
    assert cfgmod.sequence_config_from(None).frame.joints_category == "AMASS"
    with pytest.raises(TypeError):
        cfgmod.frame_config_from({"no_such_field": 1})


def test_mano_and_flame_requests_are_rejected_before_anything_is_loaded():
    """check_request (reference api/frame.py:60-75 accepts them; this engine's scope does not): NotImplementedError up
    front, not a failure deep inside the fit."""
    from keypoints2body_amd.api import common
    from keypoints2body_amd.core.config import FrameOptimizeConfig
    for name in ("mano", "flame"):
        with pytest.raises(NotImplementedError):
            common.check_request(FrameOptimizeConfig(), name)
    for name in ("smpl", "smplh", "smplx"):
        common.check_request(FrameOptimizeConfig(), name)
    with pytest.raises(ValueError):
        common.check_request(FrameOptimizeConfig(), "nope")


def test_smplx_module_constants_are_adopted_with_expression_and_beta_split():
    """an smplx SMPLX module keeps the expression directions in ``expr_dirs``; the betas / expression split is
    the module's; a non-zero ``pose_mean`` (flat_hand_mean=False) or PCA hands are refused.  Stub object, no smplx needed."""
    from types import SimpleNamespace

    import pytest

    from keypoints2body_amd.models.body_model import smplx_constants
    V, J = 40, 55
    rng = np.random.default_rng(0)
    f = lambda *s: torch.tensor(rng.standard_normal(s), dtype=torch.float32)
    stub = SimpleNamespace(v_template=f(V, 3), shapedirs=f(V, 3, 16), expr_dirs=f(V, 3, 10), posedirs=f(9 * (J - 1), 3 * V),
                           J_regressor=f(J, V), lbs_weights=f(V, J), parents=torch.arange(-1, J - 1), num_betas=16,
                           num_expression_coeffs=10, pose_mean=torch.zeros(3 * J), use_pca=False,
                           vertex_joint_selector=SimpleNamespace(extra_joints_idxs=torch.tensor([1, 2, 3])))
    c = smplx_constants(stub)
    assert c["shapedirs"].shape == (V, 3, 26) and c["num_betas"] == 16 and c["model_type"] == "smplx"
    assert np.array_equal(c["shapedirs"][:, :, 16:], stub.expr_dirs.numpy()) and list(c["extra_vertex_ids"]) == [1, 2, 3]
    stub.num_betas = 10                                   # fewer betas than the file holds: smplx slices, so does this
    assert smplx_constants(stub)["shapedirs"].shape == (V, 3, 20)
    stub.pose_mean = torch.full((3 * J,), 0.1)
    with pytest.raises(NotImplementedError, match="flat_hand_mean"):
        smplx_constants(stub)
    stub.pose_mean, stub.use_pca = None, True
    with pytest.raises(NotImplementedError, match="use_pca"):
        smplx_constants(stub)
    # a plain SMPL module: no expr_dirs, every beta kept
    smpl = SimpleNamespace(v_template=f(V, 3), shapedirs=f(V, 3, 10), posedirs=f(207, 3 * V), J_regressor=f(24, V),
                           lbs_weights=f(V, 24), parents=torch.arange(-1, 23))
    c = smplx_constants(smpl)
    assert c["shapedirs"].shape == (V, 3, 10) and c["model_type"] == "smpl" and c["extra_vertex_ids"] is None
