"""GPU: the oracle at BASELINE.json's REAL sizes - not only size-independent properties.

* LBS (reference seam: the final forward of ``core/fitters/world_space.py:258-278``): 4096 and 10 000 SMPL frames through the
  stream kernel (its full-tile fast path, the counted ``vmcnt`` waits and the XCD walk over 32 / 79 frame groups x 54 vertex
  groups) and 1024 SMPL-X frames through the tile kernel ``<7, 6>``, random LARGE rotations; vertices and all output joints of
  >= 128 sampled frames spread over the first / middle / last frame groups against ``TorchSMPL`` / ``TorchSMPLX`` at
  5e-6 x scale.
* Fit (``world_space.py:248-256``): 64 sampled frames of the 4096-frame bench problem (paired shape) and of the 1024-frame one
  (split shape) against ``oracle.fit_torch.fit_world_adam`` on exactly those frames, 100 iterations, at the north star's 1e-4.
* Partial tiles of the stream kernel: batches with B % 128 in 1..15 and joints-only calls, where whole waves
  have no store to issue, under HBM pressure from a concurrent copy - results must equal the quiet run bit for bit and the oracle.
"""
import numpy as np
import pytest
import torch

from keypoints2body_amd import native, synthetic
from tests import helpers as H

pytestmark = pytest.mark.gpu
LBS_TOL = 5e-6


def _sample_rows(B, n=132):
    """Frames of the first, middle and last 128-frame groups + a spread over the rest."""
    g = np.random.default_rng(B)
    first, last = np.arange(0, min(44, B)), np.arange(max(0, B - 44), B)
    mid0 = (B // 2) // 128 * 128
    mid = np.arange(mid0, min(mid0 + 22, B))
    rest = g.choice(B, size=max(0, n - len(first) - len(last) - len(mid)), replace=False)
    return np.unique(np.concatenate([first, mid, last, rest]))


def _large_poses(B, dims, nb, seed):
    rng = np.random.default_rng(seed)
    go = rng.uniform(-3.0, 3.0, (B, 3)).astype(np.float32)
    pose = (0.6 * rng.standard_normal((B, dims))).astype(np.float32)
    pose[B // 3] = 0.0                                   # identity rotations in one frame
    shape = rng.uniform(-2.5, 2.5, (B, nb)).astype(np.float32)
    tr = rng.uniform(-5.0, 5.0, (B, 3)).astype(np.float32)
    return go, pose, shape, tr


@pytest.mark.parametrize("B", [4096, 10000])
def test_smpl_lbs_full_size_matches_oracle_on_sampled_frames(B):
    go, bp, be, tr = _large_poses(B, 69, 10, seed=B)
    j, v = H.native_model().lbs(H.cuda(go), H.cuda(bp), H.cuda(be), H.cuda(tr))
    rows = _sample_rows(B)
    assert len(rows) >= 128
    t = lambda a: torch.tensor(a[rows])
    with torch.no_grad():
        want = H.oracle_model()(global_orient=t(go), body_pose=t(bp), betas=t(be), transl=t(tr))
    scale = max(1.0, float(want.vertices.abs().max()))
    idx = torch.as_tensor(rows, device="cuda")
    assert (v[idx].cpu() - want.vertices).abs().max().item() < LBS_TOL * scale
    assert (j[idx].cpu() - want.joints).abs().max().item() < LBS_TOL * scale
    j2, _ = H.native_model().lbs(H.cuda(go), H.cuda(bp), H.cuda(be), H.cuda(tr), want_vertices=False)
    assert torch.equal(j, j2)                           # the joints-only call (21-vertex operand set) agrees bit for bit


def test_smplx_lbs_1024_frames_match_oracle_on_sampled_frames():
    B = 1024
    go, pose, shape, tr = _large_poses(B, 162, 20, seed=77)
    pose[:, 63:] *= 0.5                                  # (jaw, eyes, fingers: smaller but still far from the linear range)
    m = H.native_model_x()
    j, v = m.lbs(H.cuda(go), H.cuda(pose), H.cuda(shape), H.cuda(tr))
    rows = _sample_rows(B)
    cols = np.cumsum([0, 63, 3, 3, 3, 45, 45])
    names = ("body_pose", "jaw_pose", "leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose")
    kw = {n: torch.tensor(pose[rows, cols[i]:cols[i + 1]]) for i, n in enumerate(names)}
    with torch.no_grad():
        want = H.oracle_model_x()(global_orient=torch.tensor(go[rows]), betas=torch.tensor(shape[rows, :10]),
                                  expression=torch.tensor(shape[rows, 10:]), transl=torch.tensor(tr[rows]), **kw)
    scale = max(1.0, float(want.vertices.abs().max()))
    idx = torch.as_tensor(rows, device="cuda")
    assert tuple(v.shape) == (B, 10475, 3) and tuple(j.shape) == (B, 127, 3)
    assert (v[idx].cpu() - want.vertices).abs().max().item() < LBS_TOL * scale
    assert (j[idx].cpu() - want.joints).abs().max().item() < LBS_TOL * scale


@pytest.mark.parametrize("B", [1024, 4096])             # the split shape (configs[1]) and the paired shape (the headline)
def test_fit_full_size_matches_oracle_on_sampled_frames(B):
    from oracle.fit_torch import fit_world_adam
    m = H.native_model()
    p = synthetic.make_poses(B, seed=1000)
    j, _ = m.lbs(*map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl)), want_vertices=False)
    j3d = j[:, :22].contiguous()
    z = lambda c: torch.zeros(B, c, device="cuda")
    j0, _ = m.lbs(z(3), z(69), z(10), None, want_vertices=False)
    tr0 = (j3d[:, 0] - j0[:, 0]).contiguous()
    cfg = native.default_fit_config()
    cfg.num_iters = 100
    out = native.fit_world(m, H.native_prior(), cfg, list(range(22)), j3d, None, z(3), z(69), z(10), tr0)
    rows = np.unique(np.concatenate([np.arange(0, 16), np.arange(B - 16, B), np.random.default_rng(B).choice(B, 32, replace=False)]))
    idx = torch.as_tensor(rows, device="cuda")
    n = len(rows)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    ref = fit_world_adam(H.oracle_model(), H.oracle_prior(), torch.zeros(n, 3), torch.zeros(n, 69), torch.zeros(n, 10),
                         tr0[idx].cpu(), j3d[idx].cpu(), None, num_iters=100)
    worst = 0.0
    for k in ("global_orient", "body_pose", "betas", "transl"):
        dev = (out[k][idx].cpu() - getattr(ref, k)).abs().max().item()
        worst = max(worst, dev)
        assert dev < 1e-4, (k, dev)
    assert (out["loss"][idx].cpu() - ref.loss).abs().max().item() < 2e-4 * float(ref.loss.abs().max())
    print(f"full-size fit, {B} frames, {n} sampled: worst parameter deviation {worst:.2e}")


@pytest.mark.parametrize("B", [1, 3, 7, 15, 128 + 9, 256 + 15])
def test_stream_kernel_partial_tiles_under_memory_pressure(B):
    """Waves without a single valid (frame, vertex) issue no store, so the counted waits of a partial tile may count the fills
    only; a wrong count shows when HBM is busy (the all-gather is deliberately overlapped with this kernel in the product)."""
    go, bp, be, tr = _large_poses(B, 69, 10, seed=40 + B)
    m = H.native_model()
    args = (H.cuda(go), H.cuda(bp), H.cuda(be), H.cuda(tr))
    with torch.no_grad():
        want = H.oracle_model()(global_orient=torch.tensor(go), body_pose=torch.tensor(bp), betas=torch.tensor(be),
                                transl=torch.tensor(tr))
    scale = max(1.0, float(want.vertices.abs().max()))
    j_quiet, v_quiet = m.lbs(*args)
    jo_quiet, _ = m.lbs(*args, want_vertices=False)
    assert (v_quiet.cpu() - want.vertices).abs().max().item() < LBS_TOL * scale
    assert (j_quiet.cpu() - want.joints).abs().max().item() < LBS_TOL * scale
    assert torch.equal(j_quiet, jo_quiet)
    # the same calls beside a copy stream that keeps HBM saturated
    side = torch.cuda.Stream()
    src = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(12):
            dst.copy_(src, non_blocking=True)
    for _ in range(6):
        j, v = m.lbs(*args)
        jo, _ = m.lbs(*args, want_vertices=False)
        assert torch.equal(v, v_quiet) and torch.equal(j, j_quiet) and torch.equal(jo, j_quiet)
    torch.cuda.synchronize()
