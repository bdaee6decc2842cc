"""N>1 path on CPU: two gloo ranks shard a sequence, fit their blocks (the oracle stands
in for the HIP kernel here, tests being allowed to), all-gather once, and every rank must
hold exactly the single-process result in frame order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from keypoints2body_amd import parallel


def test_shard_bounds_cover_every_frame_once():
    for T in (0, 1, 5, 8, 4096, 10000):
        for G in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(T, G, r) for r in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == T
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - s for s, e in spans) == parallel.rows_per_rank(T, G) == (T + G - 1) // G
            # balanced: block sizes differ by at most one, so no rank is empty unless there are more ranks than frames
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1
            assert T < G or all(e > s for s, e in spans)
            keep = parallel.valid_rows(T, G)
            per = parallel.rows_per_rank(T, G)
            assert keep.tolist() == [r * per + i for r, (s, e) in enumerate(spans) for i in range(e - s)]


def test_single_rank_async_gather_returns_no_work():
    out = {"global_orient": torch.randn(2, 3), "body_pose": torch.randn(2, 69), "betas": torch.randn(2, 10),
           "transl": torch.randn(2, 3), "loss": torch.randn(2)}
    packed, work = parallel.gather_fit_outputs(out, None, async_op=True)
    assert work is None and torch.equal(packed, parallel.pack_outputs(out))


def test_pack_unpack_roundtrip():
    out = {"global_orient": torch.randn(4, 3), "body_pose": torch.randn(4, 69), "betas": torch.randn(4, 10),
           "transl": torch.randn(4, 3), "loss": torch.randn(4)}
    back = parallel.unpack_outputs(parallel.pack_outputs(out), 10, 69)
    assert all(torch.equal(out[k], back[k]) for k in out)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fit_block(sl):
    from oracle.fit_torch import fit_world_adam
    from tests import helpers as H
    d = H.load_case("amass_noisy_conf")
    t = lambda k: torch.tensor(d[k][:5][sl])
    if t("j3d").shape[0] == 0:
        e = lambda c: torch.zeros(0, c)
        return {"global_orient": e(3), "body_pose": e(69), "betas": e(10), "transl": e(3), "loss": torch.zeros(0)}
    o = fit_world_adam(H.oracle_model(), H.oracle_prior(), t("init_global_orient"), t("init_body_pose"),
                       t("init_betas"), t("init_transl"), t("j3d"), torch.tensor(d["conf"]), num_iters=3)
    return {"global_orient": o.global_orient, "body_pose": o.body_pose, "betas": o.betas, "transl": o.transl,
            "loss": o.loss}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = parallel.fit_frames_sharded(_fit_block, 5, 10, 69, dist)     # 5 frames over 2 ranks: 3 + 2
        # the asynchronous form used by bench.py (collective enqueued, other work launched, then waited for)
        mine = _fit_block(slice(*parallel.shard_bounds(5, world, rank)))
        gathered, work = parallel.gather_fit_outputs(mine, dist, pad_to=3, async_op=True)
        assert work is not None
        work.wait()
        sync = parallel.gather_fit_outputs(mine, dist, pad_to=3)
        assert torch.equal(gathered, sync)
        q.put((rank, {k: v.numpy() for k, v in res.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_fit_equals_single_process():
    torch.set_num_threads(2)
    single = _fit_block(slice(0, 5))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in (0, 1):
        for k, v in single.items():
            assert got[rank][k].shape == tuple(v.shape)
            assert np.abs(got[rank][k] - v.numpy()).max() < 2e-6, (rank, k)


def _shape_terms_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7 + rank)
        loss, gb, gt = torch.rand((), generator=g), torch.randn(10, generator=g), torch.randn(3, generator=g)
        rl, rb, rt = parallel.allreduce_shape_terms(loss, gb, gt, dist)
        q.put((rank, rl.item(), rb.numpy(), rt.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_shape_pass_terms_all_reduce_to_the_same_sum_on_every_rank():
    """The shape pre-pass shards its frames and sums loss / d beta / d transl with ONE all-reduce per closure call."""
    one = parallel.allreduce_shape_terms(torch.tensor(1.5), torch.ones(10), torch.ones(3), None)
    assert float(one[0]) == 1.5 and one[1].shape == (10,) and one[2].shape == (3,)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shape_terms_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=100) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    want = [0.0, np.zeros(10, np.float32), np.zeros(3, np.float32)]
    for r in range(2):
        g = torch.Generator().manual_seed(7 + r)
        want[0] += torch.rand((), generator=g).item(); want[1] += torch.randn(10, generator=g).numpy(); want[2] += torch.randn(3, generator=g).numpy()
    for rank, l, b, t in got:
        assert abs(l - want[0]) < 1e-6 and np.abs(b - want[1]).max() < 1e-6 and np.abs(t - want[2]).max() < 1e-6
    assert got[0][1] == got[1][1] and np.array_equal(got[0][2], got[1][2]) and np.array_equal(got[0][3], got[1][3])
