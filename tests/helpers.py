"""Shared test helpers: synthetic assets, golden loading, oracle objects (CPU) and
native handles (GPU).  Tests are the only place (besides smoke() and bench's
cpu_baseline leg) allowed to touch ``oracle/``."""
from __future__ import annotations

import functools
from pathlib import Path

import numpy as np
import torch

from keypoints2body_amd import synthetic

GOLDEN = Path(__file__).resolve().parent / "golden"
WORLD_CASES = ("amass_zero_init", "amass_batched", "amass_noisy_conf", "amass_followup",
               "amass_freeze_betas", "smpl24_zero_init", "generic_indices")
CATEGORY_INDEX = {"AMASS": list(range(22)), "SMPL24": list(range(24))}


@functools.lru_cache(maxsize=None)
def body_consts(seed: int = 0):
    return synthetic.make_body_model(seed)


@functools.lru_cache(maxsize=None)
def oracle_model(seed: int = 0, double: bool = False):
    from oracle.smpl_torch import TorchSMPL
    return TorchSMPL(body_consts(seed), dtype=torch.float64 if double else torch.float32)


@functools.lru_cache(maxsize=None)
def body_consts_x(seed: int = 0):
    return synthetic.make_body_model_x(seed)


@functools.lru_cache(maxsize=None)
def oracle_model_x(seed: int = 0, double: bool = False):
    from oracle.smpl_torch import TorchSMPLX
    return TorchSMPLX(body_consts_x(seed), dtype=torch.float64 if double else torch.float32)


def body_consts_h(seed: int = 0):
    return synthetic.make_body_model_h(seed)


@functools.lru_cache(maxsize=None)
def oracle_model_h(seed: int = 0):
    from oracle.smpl_torch import TorchSMPLH
    return TorchSMPLH(body_consts_h(seed))


@functools.lru_cache(maxsize=None)
def native_model_h(seed: int = 0):
    from keypoints2body_amd.native import NativeModel
    c = body_consts_h(seed)
    return NativeModel(c.v_template, c.shapedirs, c.posedirs, c.J_regressor, c.lbs_weights, c.parents,
                       c.extra_vertex_ids)


def load_smplh_case(name: str):
    return dict(np.load(GOLDEN / f"smplh_fit_{name}.npz"))


def load_smplx_case(name: str):
    return dict(np.load(GOLDEN / f"smplx_fit_{name}.npz"))


@functools.lru_cache(maxsize=None)
def native_model_x(seed: int = 0):
    from keypoints2body_amd.native import NativeModel
    c = body_consts_x(seed)
    return NativeModel(c.v_template, c.shapedirs, c.posedirs, c.J_regressor, c.lbs_weights, c.parents,
                       c.extra_vertex_ids)


@functools.lru_cache(maxsize=None)
def gmm_fixture():
    return dict(np.load(GOLDEN / "gmm_synth.npz"))


@functools.lru_cache(maxsize=None)
def oracle_prior():
    """Oracle prior carrying exactly the buffers the reference derived (fixture)."""
    from oracle.fit_torch import GMMPrior
    g = gmm_fixture()
    p = GMMPrior(g["means"], g["covars"].astype(np.float64), g["weights"])
    p.means = torch.tensor(g["ref_means"])
    p.precisions = torch.tensor(g["ref_precisions"])
    p.nll_weights = torch.tensor(g["ref_nll_weights"])
    return p


def load_case(name: str):
    return dict(np.load(GOLDEN / f"world_fit_{name}.npz"))


def case_indices(d):
    """(model joint index per target) for a golden case."""
    cat = str(d["category"])
    if cat == "GENERIC":
        return [int(i) for i in d["target_model_indices"]]
    return CATEGORY_INDEX[cat]


@functools.lru_cache(maxsize=None)
def native_model(seed: int = 0):
    from keypoints2body_amd.native import NativeModel
    c = body_consts(seed)
    return NativeModel(c.v_template, c.shapedirs, c.posedirs, c.J_regressor, c.lbs_weights, c.parents,
                       c.extra_vertex_ids)


@functools.lru_cache(maxsize=None)
def native_prior():
    from keypoints2body_amd.native import NativePrior
    g = gmm_fixture()
    return NativePrior(g["ref_means"], g["ref_precisions"], g["ref_nll_weights"])


def cuda(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32).cuda().contiguous()


LAUNCH_SHAPES = {"auto": 0, "split": 1, "split_paired": 2, "paired": 3, "wide": 4}     # k2b_fit_config.debug_launch_shape


def native_fit(d, num_iters=None, want_grad=False, rows=None, shape="auto"):
    """Run the HIP fit on the inputs of golden case ``d`` (``shape`` forces a launch shape of the fused kernel)."""
    from keypoints2body_amd import native
    cfg = native.default_fit_config()
    cfg.debug_launch_shape = LAUNCH_SHAPES[shape]
    cfg.num_iters = int(d["num_iters"]) if num_iters is None else int(num_iters)
    cfg.pose_preserve_weight = 5.0 if int(d["seq_ind"]) > 0 else 0.0
    cfg.freeze_betas = int(d["freeze_betas"])
    sl = slice(None) if rows is None else rows
    conf = cuda(d["conf"]) if int(d["has_conf"]) else None
    return native.fit_world(native_model(), native_prior(), cfg, case_indices(d), cuda(d["j3d"][sl]), conf,
                            cuda(d["init_global_orient"][sl]), cuda(d["init_body_pose"][sl]),
                            cuda(d["init_betas"][sl]), cuda(d["init_transl"][sl]), want_grad=want_grad)


CHAIN_CASES = ("chain_motion1_30_10", "chain_motion2_30_10", "chain_motion1_100_50")


def load_chain_case(name: str):
    """Sequence-loop golden of ``oracle/gen_golden_chain.py``: the REAL reference fitter walked through its own frame
    loop (api/sequence.py:214-281: fix_foot confidences, seq_ind = idx, prev = res.params) over real demo motion."""
    return dict(np.load(GOLDEN / f"{name}.npz"))
