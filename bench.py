#!/usr/bin/env python3
"""Benchmark of the hot path: SMPL frames fitted per second (100 Adam iterations,
22-joint AMASS targets) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W [--frames F] [--iters I]

One "step" = one pass of the hot path over one batch of F synthetic frames per GPU:
the fused fit kernel (all Adam iterations), the final LBS forward (joints + 6890
vertices per frame) and, for N > 1, the all-gather of the fitted parameters over
RCCL.  Frames shard across ranks with no data-path collective (weak scaling: F frames
per GPU).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE
JSON line; see DESIGN.md "Measurement" for the definitions of `roofline` and
`cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

# algorithmic work per unit (SURVEY.md §8d / DESIGN.md)
FIT_FLOP_PER_FRAME_ITER = 0.11e6        # analytic forward + backward + Adam, GMM prior dominates
LBS_BYTES_PER_FRAME = 83_560            # 340 B params in + 540 B joints + 82 680 B vertices out
LBS_FLOP_PER_FRAME = 13.1e6
FP32_PEAK_TFLOPS = 157.3                # MI355X_MICROARCH.md: fp32 vector = fp32-input MFMA peak
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=1024, help="frames per GPU (BASELINE configs[1]: 1024)")
    ap.add_argument("--iters", type=int, default=100, help="Adam iterations per frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=6, help="frames of the CPU baseline's B=1 loop")
    return ap.parse_args()


def build_problem(frames, seed, device):
    """Synthetic model, prior, targets and initialisation, all resident on `device`."""
    from keypoints2body_amd import synthetic
    from keypoints2body_amd.core.fitters.world_space import WorldSpaceFitter, guess_init_transl_from_root
    from keypoints2body_amd.models.body_model import BodyModel
    from keypoints2body_amd.models.smpl_data import SMPLData
    from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers

    model = BodyModel.synthetic(seed=0, device=device)
    gmm = synthetic.make_gmm(seed=0)
    prior = MaxMixturePrior(MixtureBuffers.from_mixture(gmm.means, gmm.covars, gmm.weights), device=device)
    poses = synthetic.make_poses(frames, seed=seed)
    dev = lambda a: torch.as_tensor(a, dtype=torch.float32, device=device).contiguous()
    gt = model(global_orient=dev(poses.global_orient), body_pose=dev(poses.body_pose), betas=dev(poses.betas),
               transl=dev(poses.transl), return_verts=False)
    j3d = gt.joints[:, :22].contiguous()
    zeros = lambda c: torch.zeros((frames, c), dtype=torch.float32, device=device)
    transl0 = guess_init_transl_from_root(model, zeros(72), zeros(10), j3d, joints_category="AMASS")
    init = SMPLData(betas=zeros(10), global_orient=zeros(3), body_pose=zeros(69), transl=transl0.contiguous())
    return model, prior, j3d, init, poses


def cpu_baseline(iters, n_loop_frames):
    """Reference-style CPU loop (oracle port) on this host's cores: per-frame B=1
    fits, exactly what `optimize_params_frame` does per frame, plus one batched call."""
    from keypoints2body_amd import synthetic
    from oracle.fit_torch import GMMPrior, fit_world_adam, guess_init_transl
    from oracle.smpl_torch import TorchSMPL

    # the GPU box gives one GPU's share of the host (16 CPUs); os.cpu_count() reports the whole machine
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    model = TorchSMPL(synthetic.make_body_model(0))
    g = synthetic.make_gmm(0)
    prior = GMMPrior(g.means, g.covars, g.weights)
    nb = 32
    poses = synthetic.make_poses(max(n_loop_frames, nb), seed=1000)
    t = lambda a: torch.tensor(a)
    with torch.no_grad():
        j3d = model(global_orient=t(poses.global_orient), body_pose=t(poses.body_pose), betas=t(poses.betas),
                    transl=t(poses.transl)).joints[:, :22].clone()
    B = j3d.shape[0]
    tr0 = guess_init_transl(model, torch.zeros(B, 72), torch.zeros(B, 10), j3d)
    z = lambda n, c: torch.zeros(n, c)

    def fit(sl):
        n = j3d[sl].shape[0]
        return fit_world_adam(model, prior, z(n, 3), z(n, 69), z(n, 10), tr0[sl], j3d[sl], None, num_iters=iters)

    fit(slice(0, 1))                                     # warm-up
    t0 = time.perf_counter()
    for i in range(n_loop_frames):
        fit(slice(i, i + 1))
    loop_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    fit(slice(0, nb))
    batch_s = time.perf_counter() - t0
    return {
        "value": round(n_loop_frames / loop_s, 3),
        "unit": "frames/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": (f"{n_loop_frames} frames fitted one at a time (B=1, {iters} Adam iters, full SMPL forward incl. 6890 "
                   f"vertices every iteration, torch autograd + torch.optim.Adam) in {loop_s:.2f}s; one batched "
                   f"B={nb} call: {nb / batch_s:.2f} frames/s"),
        "batched_value": round(nb / batch_s, 3),
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    # K2B_BENCH_BACKEND=gloo is a REHEARSAL switch (tools/README.md): it lets the N > 1 code path run with several
    # ranks sharing the one card of a development box, where RCCL refuses duplicate devices.  Numbers from it mean nothing.
    backend = os.environ.get("K2B_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    from keypoints2body_amd.core.fitters.world_space import WorldSpaceFitter
    from keypoints2body_amd.parallel import gather_fit_outputs

    F = args.frames
    model, prior, j3d, init, _ = build_problem(F, seed=1000 + rank, device=device)
    fitter = WorldSpaceFitter(model, step_size=1e-2, num_iters_first=args.iters, num_iters_followup=args.iters,
                              use_lbfgs=False, joints_category="AMASS", device=device, pose_prior=prior)

    ev = lambda: torch.cuda.Event(enable_timing=True)
    fit_ms, lbs_ms = [], []

    def step(record=False):
        from keypoints2body_amd import native
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        cfg = fitter._config(0, 600.0, 5.0, False, False)
        out = native.fit_world(model.native, prior.native, cfg, list(range(22)), j3d, None, init.global_orient,
                               init.body_pose, init.betas, init.transl)
        e1.record()
        # the parameter exchange is enqueued behind the fit and runs on RCCL's stream under the LBS launches
        gathered, work = gather_fit_outputs(out, dist, async_op=True) if dist is not None else (None, None)
        joints, verts = model.native.lbs(out["global_orient"], out["body_pose"], out["betas"], out["transl"])
        e2.record()
        if work is not None:
            work.wait()
        if record:
            fit_ms.append((e0, e1))
            lbs_ms.append((e1, e2))
        return out, joints, verts, gathered

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, joints, verts, gathered = step(record=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    fit_avg = float(np.mean([a.elapsed_time(b) for a, b in fit_ms]))      # ms, HIP events on the launch stream
    lbs_avg = float(np.mean([a.elapsed_time(b) for a, b in lbs_ms]))
    err_cm = float((joints[:, :22] - j3d).norm(dim=-1).mean().item() * 100)

    if rank == 0:
        total_frames = F * world * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        fit_tflops = FIT_FLOP_PER_FRAME_ITER * args.iters * F / (fit_avg * 1e-3) / 1e12
        lbs_gbs = LBS_BYTES_PER_FRAME * F / (lbs_avg * 1e-3) / 1e9
        # HBM bytes per launch measured with rocprofv3 PMC passes (profiles/README.md); only known for
        # the frame counts that were profiled
        traffic = traffic_lbs = None
        tfile = REPO / "profiles" / "traffic_r01.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                traffic, traffic_lbs = tj.get(f"fit_frames_{F}"), tj.get(f"lbs_frames_{F}")
            except Exception:
                pass
        line = {
            "metric": "SMPL frames fitted/sec (100 Adam iters, 22-joint AMASS)",
            "value": round(total_frames / elapsed, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{F} synthetic 22-joint AMASS frames per GPU, SMPL-shaped model (V=6890, J=24, 10 betas), "
                            f"{args.iters} Adam iters, world mode, final joints+vertices produced",
                "frames_per_gpu": F, "adam_iters": args.iters, "parallelism": f"frames sharded x{world}",
            },
            "roofline": {
                "kernel": "k2b_fit_world_kernel", "bound": "mfma", "achieved": round(fit_tflops, 3),
                "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fit_tflops / FP32_PEAK_TFLOPS, 4),
                "traffic": traffic, "avg_launch_ms": round(fit_avg, 4),
            },
            "roofline_lbs": {
                "kernel": "k2b_pose_setup_kernel+k2b_lbs_mfma_kernel+k2b_gather_joints_kernel", "bound": "hbm",
                "achieved": round(lbs_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(lbs_gbs / HBM_PEAK_GBS, 4), "traffic": traffic_lbs, "avg_launch_ms": round(lbs_avg, 4),
                "achieved_tflops": round(LBS_FLOP_PER_FRAME * F / (lbs_avg * 1e-3) / 1e12, 2),
            },
            "quality": {"mean_joint_error_cm": round(err_cm, 3), "mean_final_loss": round(float(out["loss"].mean()), 2)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.iters, args.cpu_frames)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
