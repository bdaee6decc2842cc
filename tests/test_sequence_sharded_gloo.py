"""The PUBLIC ``optimize_params_sequence`` under an initialised process group (two gloo ranks on CPU):
with ``use_previous_frame_init=False`` frames 1..T-1 must be sharded over the ranks and every rank must
return the full list, equal to the single-process result in frame order.

The HIP engine cannot run here, so (tests being allowed to) the oracle stands in for the kernel behind
the engine seam: a stand-in ``OptimizeEngine`` whose estimator runs ``oracle.fit_torch.fit_world_adam``.
Everything above that seam - normalisation, the frame-0 / frames-1.. split, the sharding, the gather and
the assembly of results - is the product code of ``keypoints2body_amd/api/sequence.py``.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

T_FRAMES = 6
CALLS = []          # (rank-local) sizes of the batched fits the stand-in estimator was asked for


class _Fitter:
    def __init__(self, model):
        self.model = model

    def fit_batch(self, *a, **k):       # presence selects the batched branch of the sequence API
        raise AssertionError("reached through the estimator")

    def final_forward(self, out, want_vertices=True):
        with torch.no_grad():
            o = self.model(global_orient=out["global_orient"], body_pose=out["body_pose"], betas=out["betas"],
                           transl=out["transl"])
        return o.joints, o.vertices

    def result_params(self, out, init, rows=None):
        from keypoints2body_amd.models.smpl_data import SMPLData
        sl = slice(None) if rows is None else rows
        return SMPLData(betas=out["betas"][sl], global_orient=out["global_orient"][sl], body_pose=out["body_pose"][sl],
                        transl=out["transl"][sl])


class _Estimator:
    def __init__(self, model, prior, cfg):
        self.model, self.prior, self.cfg = model, prior, cfg
        self.fitter = _Fitter(model)

    def _fit(self, init, j3d, conf, seq_ind):
        from oracle.fit_torch import fit_world_adam
        iters = self.cfg.num_iters_first if seq_ind == 0 else self.cfg.num_iters_followup
        n = j3d.shape[0]
        rep = lambda x: torch.as_tensor(x).expand(n, -1).contiguous()
        c = conf if conf is None or conf.dim() == 1 else conf[0]
        return fit_world_adam(self.model, self.prior, rep(init.global_orient), rep(init.body_pose), rep(init.betas),
                              rep(init.transl), j3d, c, num_iters=iters, seq_ind=seq_ind)

    def fit_batch(self, init, j3d, conf, seq_ind, target_model_indices=None, per_frame_conf=False, run_forward=True):
        CALLS.append(int(j3d.shape[0]))
        o = self._fit(init, j3d, conf, seq_ind)
        out = {"global_orient": o.global_orient, "body_pose": o.body_pose, "betas": o.betas, "transl": o.transl,
               "loss": o.loss}
        if not run_forward:
            return out, None, None, o.loss
        return out, o.joints, o.vertices, o.loss


class _Engine:
    def __init__(self, model, frame_config, device=None, model_type="smpl", pose_prior=None):
        self.estimator = _Estimator(model, pose_prior, frame_config)

    def fit_frame(self, init_params, j3d, conf_3d, seq_ind, target_model_indices=None):
        from keypoints2body_amd.models.smpl_data import BodyModelFitResult, SMPLData
        o = self.estimator._fit(init_params, j3d, conf_3d, seq_ind)
        return BodyModelFitResult(params=SMPLData(betas=o.betas, global_orient=o.global_orient, body_pose=o.body_pose,
                                                  transl=o.transl), vertices=o.vertices, joints=o.joints,
                                  loss=o.loss.sum())


def _run_public_api():
    """optimize_params_sequence on the first T_FRAMES frames of a golden case, oracle behind the engine seam."""
    from keypoints2body_amd.api import common, sequence
    from keypoints2body_amd.core.config import FrameOptimizeConfig, SequenceOptimizeConfig
    from keypoints2body_amd.models.smpl_data import SMPLData
    from tests import helpers as H

    common.resolve_device = lambda device: torch.device("cpu")
    common.obtain_model = lambda model, body_model, device: model
    sequence.OptimizeEngine = _Engine
    d = H.load_case("amass_batched")
    j3d = torch.tensor(np.tile(d["j3d"], (3, 1, 1))[:T_FRAMES])
    j3d = j3d + 0.01 * torch.arange(T_FRAMES).view(-1, 1, 1)            # distinct frames
    t = lambda k: torch.tensor(d[k][:1])
    init = SMPLData(betas=t("init_betas"), global_orient=t("init_global_orient"), body_pose=t("init_body_pose"),
                    transl=t("init_transl"))
    cfg = SequenceOptimizeConfig(
        frame=FrameOptimizeConfig(use_lbfgs=False, num_iters_first=3, num_iters_followup=2, joints_category="AMASS"),
        use_previous_frame_init=False, use_shape_optimization=False)
    res = sequence.optimize_params_sequence(j3d, init_params=init, body_model="smpl", joint_layout="AMASS",
                                            model=H.oracle_model(), config=cfg, pose_prior=H.oracle_prior(),
                                            mean_params=(torch.zeros(1, 72), torch.zeros(1, 10)))
    pack = lambda r: np.concatenate([r.params.global_orient.numpy().ravel(), r.params.body_pose.numpy().ravel(),
                                     r.params.betas.numpy().ravel(), r.params.transl.numpy().ravel(),
                                     np.asarray(r.loss, dtype=np.float32).reshape(-1),
                                     r.joints.numpy().ravel(), r.vertices.numpy().ravel()[:300]])
    return np.stack([pack(r) for r in res])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        got = _run_public_api()
        q.put((rank, got, list(CALLS)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_public_sequence_api_shards_independent_frames_over_two_ranks():
    torch.set_num_threads(2)
    CALLS.clear()
    single = _run_public_api()
    assert single.shape[0] == T_FRAMES and CALLS == [T_FRAMES - 1]       # one batched fit of frames 1..T-1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        rank, arr, calls = q.get(timeout=500)
        got[rank] = (arr, calls)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # 5 remaining frames over 2 ranks: blocks of 3 and 2, each rank fitted only its own block
    assert got[0][1] == [3] and got[1][1] == [2]
    for rank in (0, 1):
        assert got[rank][0].shape == single.shape
        assert np.abs(got[rank][0] - single).max() < 5e-6, rank
    assert np.array_equal(got[0][0][:, :85], got[1][0][:, :85])          # gathered parameters: same bits on every rank
