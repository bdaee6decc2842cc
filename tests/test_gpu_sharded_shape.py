"""Shape pre-pass sharded over two ranks that share the one card of the test box (gloo rendezvous and
all-reduce: RCCL refuses two ranks on one device; the kernels, the sharding and the reduction are the real
ones).  Both ranks must return the betas of the single-process pass, which the golden pins to the reference."""
import os
import socket

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from keypoints2body_amd.core.config import SequenceOptimizeConfig
    from keypoints2body_amd.core.engine import optimize_shape_pass
    from keypoints2body_amd.models.body_model import BodyModel
    from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        g = H.gmm_fixture()
        prior = MaxMixturePrior(MixtureBuffers(g["ref_means"], g["ref_precisions"], g["ref_nll_weights"].reshape(-1)))
        model = BodyModel.synthetic(0)
        d = dict(np.load(H.GOLDEN / "shape_pass.npz"))
        cfg = SequenceOptimizeConfig(num_shape_frames=int(d["num_shape_frames"]), num_shape_iters=int(d["num_shape_iters"]))
        betas = optimize_shape_pass(model, cfg, torch.tensor(d["init_betas"]), torch.tensor(d["mean_pose"]),
                                    torch.tensor(d["j3d"]), torch.tensor(d["conf"]), model.device, pose_prior=prior, dist=dist)
        q.put((rank, betas.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_shape_pass_matches_single_process_and_golden(world):
    import torch.multiprocessing as mp
    d = dict(np.load(H.GOLDEN / "shape_pass.npz"))
    assert int(d["num_shape_frames"]) == 4            # world 3 -> blocks of 2, 2, 0 frames: an empty trailing shard
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        assert np.abs(got[r] - d["out_betas"]).max() < 1e-4, r
        assert np.array_equal(got[r], got[0])          # every rank holds the same bits
