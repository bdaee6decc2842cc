"""Multi-GPU execution of the fitting path: frames of a sequence shard across the
GPUs of one node, one process per GPU, no communication during the Adam iterations,
ONE collective at the end.

The reference has no distributed code at all; what shards is its frame loop
(reference ``keypoints2body/api/sequence.py:214-281``) in the modes where frames are
independent: per-frame fits as ``optimize_params_frame`` does them (``api/frame.py:213-219``,
always ``seq_ind=0``) or a sequence with ``use_previous_frame_init=False``.

(The optional shape pre-pass, which couples the frames through the shared betas, shards the same way with one
all-reduce of NB + 4 floats per closure evaluation: ``allreduce_shape_terms``.)

Collective: a single all-gather of the packed fitted parameters (+ per-frame loss):
``3J + NB + 3 + 1`` floats per frame (344 B for SMPL) — latency-bound on the fully
connected xGMI mesh, so one direct ``all_gather_into_tensor`` (RCCL), never a ring
pipeline of small buckets.  Vertices and joints stay sharded on the GPU that
produced them.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch

PARAM_KEYS = ("global_orient", "body_pose", "betas", "transl")


def shard_bounds(num_frames: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [start, stop) of `rank`, balanced: the first T mod G ranks own ceil(T / G) frames, the others
    floor(T / G) - so no rank is empty unless T < G (the ceil-only rule left trailing ranks without frames already at
    T = 4 on 3 ranks).  Every block is at most ``rows_per_rank`` long, which is what the exchanges pad to."""
    base, extra = divmod(num_frames, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def rows_per_rank(num_frames: int, world_size: int) -> int:
    """Rows every rank contributes to an all-gather of per-frame tensors (the longest block; shorter ones are zero-padded)."""
    return (num_frames + world_size - 1) // world_size


def pack_outputs(out: Dict[str, torch.Tensor]) -> torch.Tensor:
    """(B, P+1) row per frame: [global_orient | body_pose | betas | transl | loss]."""
    return torch.cat([out[k] for k in PARAM_KEYS] + [out["loss"].reshape(-1, 1)], dim=1).contiguous()


def unpack_outputs(packed: torch.Tensor, num_betas: int, pose_dim: int) -> Dict[str, torch.Tensor]:
    sizes = (3, pose_dim, num_betas, 3, 1)
    go, bp, be, tr, loss = torch.split(packed, sizes, dim=1)
    return {"global_orient": go.contiguous(), "body_pose": bp.contiguous(), "betas": be.contiguous(),
            "transl": tr.contiguous(), "loss": loss.reshape(-1).contiguous()}


def gather_fit_outputs(out: Dict[str, torch.Tensor], dist=None, pad_to: Optional[int] = None, async_op: bool = False):
    """All-gather the packed outputs of every rank: returns (world * rows, P+1) on every rank.

    `dist` is the ``torch.distributed`` module (process group already initialised; backend
    ``nccl`` = RCCL on the GPUs, ``gloo`` in the CPU tests).  `pad_to` = common row count
    when shards are uneven (rows beyond a rank's own frames are zero).

    ``async_op=True`` returns ``(tensor, work)``: the collective is only enqueued (RCCL runs it on its
    own stream, ordered after everything already launched on the current one), so the caller can launch
    the LBS forward behind it and call ``work.wait()`` afterwards - the 340 B/frame exchange then travels
    over xGMI while the vertex kernel runs.  ``work`` is ``None`` for a single rank.
    """
    packed = pack_outputs(out)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return (packed, None) if async_op else packed
    rows = packed.shape[0] if pad_to is None else pad_to
    if packed.shape[0] != rows:
        padded = packed.new_zeros((rows, packed.shape[1]))
        padded[: packed.shape[0]] = packed
        packed = padded
    world = dist.get_world_size()
    gathered = packed.new_empty((world * rows, packed.shape[1]))
    work = dist.all_gather_into_tensor(gathered, packed, async_op=async_op)
    return (gathered, work) if async_op else gathered


def _world(dist) -> Tuple[int, int]:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return 1, 0
    return dist.get_world_size(), dist.get_rank()


def gather_rows(t: torch.Tensor, dist, pad_to: int, async_op: bool = False):
    """All-gather of one per-frame tensor (rows = this rank's frames, zero-padded to `pad_to` rows so that uneven and
    empty shards take part): returns ``(world * pad_to, ...)`` on every rank, and the work handle with ``async_op``."""
    rows = t.shape[0]
    if rows != pad_to:
        padded = t.new_zeros((pad_to,) + tuple(t.shape[1:]))
        padded[:rows] = t
        t = padded
    t = t.contiguous()
    gathered = t.new_empty((dist.get_world_size() * pad_to,) + tuple(t.shape[1:]))
    work = dist.all_gather_into_tensor(gathered, t, async_op=async_op)
    return (gathered, work) if async_op else gathered


def valid_rows(num_frames: int, world: int, device=None) -> torch.Tensor:
    """Indices of the real frames inside a gathered ``(world * ceil(T / world), ...)`` tensor (padding rows dropped)."""
    per = rows_per_rank(num_frames, world)
    idx = [torch.arange(r * per, r * per + (shard_bounds(num_frames, world, r)[1] - shard_bounds(num_frames, world, r)[0]))
           for r in range(world)]
    return torch.cat(idx).to(device) if idx else torch.zeros((0,), dtype=torch.long, device=device)


def _rows_tensor(t, name: str) -> torch.Tensor:
    if t is None:
        raise RuntimeError(f"fit_forward_exchange: the forward returned no {name} although their exchange was requested "
                           "(an empty block must return a zero-row tensor, so that every rank enters the collective)")
    return t


def fit_forward_exchange(fit_fn: Callable[[], Dict[str, torch.Tensor]], forward_fn, dist=None, pad_to: Optional[int] = None,
                         gather_joints: bool = True, gather_vertices: bool = False) -> Dict[str, object]:
    """ONE rank's share of a frame-sharded fit - what both ``optimize_params_sequence`` (independent frames under a process
    group) and ``bench.py`` run per step, in the order that hides the exchange:

      1. ``fit_fn()`` fits THIS rank's block (no communication during the Adam iterations);
      2. the all-gather of the packed parameters + per-frame loss (344 B/frame for SMPL) is only ENQUEUED - RCCL runs it on
         its own stream behind the fit kernel;
      3. ``forward_fn(out)`` = the final forward over the rank's OWN block only (joints + vertices stay sharded);
      4. the joints of the block (540 B/frame) are all-gathered; the vertices (82.7 KB/frame) only with ``gather_vertices``
         (a second, separately reported exchange: SURVEY §8e), otherwise every rank keeps the vertices of its own block;
      5. wait for the collectives (stream-ordered: the host is not blocked).

    `pad_to` = rows every rank contributes (ceil(T / world); shards may be short or - with more ranks than frames - empty).
    Returns a dict: ``local`` (this rank's fit outputs), ``packed`` (world * pad_to, P + 1) or None for one rank, ``joints``
    (gathered, or local), ``vertices`` (local block, or gathered), ``vertices_gathered``.

    Which collectives run is decided by the ARGUMENTS alone (`gather_joints`, `gather_vertices`), never by what this rank's
    block holds: a rank without frames fits and forwards ZERO rows (`fit_fn` / `forward_fn` must return zero-row tensors of the
    usual trailing shapes then, not None) and enters every all-gather with padding only.  A `forward_fn` that returns None
    for a tensor an exchange was asked for is a programming error and raises BEFORE the first collective is entered only if
    every rank sees it; so do not make that depend on the data.
    """
    out = fit_fn()
    world, _ = _world(dist)
    if world == 1:
        joints, verts = forward_fn(out)
        return {"local": out, "packed": None, "joints": joints, "vertices": verts, "vertices_gathered": True}
    rows = pad_to if pad_to is not None else out["loss"].shape[0]
    packed, work = gather_fit_outputs(out, dist, pad_to=rows, async_op=True)
    joints, verts = forward_fn(out)
    works = [work]
    if gather_joints:
        joints, w = gather_rows(_rows_tensor(joints, "joints"), dist, rows, async_op=True)
        works.append(w)
    gathered_v = False
    if gather_vertices:
        verts, w = gather_rows(_rows_tensor(verts, "vertices"), dist, rows, async_op=True)
        works.append(w)
        gathered_v = True
    for w in works:
        if w is not None:
            w.wait()
    return {"local": out, "packed": packed, "joints": joints, "vertices": verts, "vertices_gathered": gathered_v}


def fit_frames_sharded(fit_fn: Callable[[slice], Dict[str, torch.Tensor]], num_frames: int, num_betas: int,
                       pose_dim: int, dist=None) -> Dict[str, torch.Tensor]:
    """Fit `num_frames` independent frames across all ranks and return the concatenated
    parameters (+ per-frame loss) of the whole sequence on every rank.

    `fit_fn(frame_slice)` fits this rank's block and returns the usual dict of (b, .)
    tensors; on the GPUs it wraps ``WorldSpaceFitter.fit_batch``.  Frame order is preserved.
    """
    world, rank = _world(dist)
    start, stop = shard_bounds(num_frames, world, rank)
    per = rows_per_rank(num_frames, world)
    out = fit_fn(slice(start, stop))
    gathered = gather_fit_outputs(out, dist if world > 1 else None, pad_to=per)
    if world > 1:
        gathered = gathered.index_select(0, valid_rows(num_frames, world, gathered.device))   # drop the padding rows
    return unpack_outputs(gathered, num_betas, pose_dim)


def allreduce_shape_terms(loss: torch.Tensor, g_beta: torch.Tensor, g_transl: torch.Tensor, dist=None):
    """Sum over ranks of the shape pre-pass's per-shard terms (reference ``core/shape.py:75-101`` sums them over
    all frames): scalar loss, d/d betas (NB,), summed d/d transl (3,).  One all-reduce of NB + 4 floats; every rank
    receives the same bits, so the L-BFGS iterations that follow stay identical on all ranks with no broadcast."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return loss, g_beta, g_transl
    nb = g_beta.numel()
    buf = torch.cat([g_beta.reshape(-1), g_transl.reshape(-1), loss.reshape(1)]).contiguous()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf[nb + 3], buf[:nb], buf[nb:nb + 3]
