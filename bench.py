#!/usr/bin/env python3
"""Benchmark of the hot path: SMPL frames fitted per second (100 Adam iterations,
22-joint AMASS targets) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W [--total-frames T | --frames F] [--iters I] [--model smpl|smplx]

``--gpus N > 1`` without a launcher (no WORLD_SIZE in the environment) starts its own N ranks with
``python -m torch.distributed.run`` before touching the GPU and relays rank 0's line; under a launcher
(the driver's ``torch.distributed.run ... bench.py --gpus N``) the process is one rank.

Default workload = the north-star headline: ONE synthetic AMASS sequence of T = 4096 frames whose
frames shard over the N ranks in contiguous blocks (``parallel.shard_bounds``; strong scaling: the
total work is fixed, N = 1 fits all 4096 frames on one GPU).  ``--frames F`` switches to weak scaling
(F frames per GPU).  One "step" = one pass of the hot path over the rank's frames =
``parallel.fit_forward_exchange``, the function the public ``optimize_params_sequence`` runs for
independent frames: the fused fit kernel (all Adam iterations), the all-gather of the fitted parameters
over RCCL (N > 1: enqueued behind the fit, on RCCL's stream), the final LBS forward over the rank's OWN
block (joints + all vertices per frame), the all-gather of the joints, the wait for both.  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line; BASELINE configs[1] (1024 frames
per GPU), configs[2] (10 000-frame sequence, sharded) and configs[3] (SMPL-X, 1024 frames per GPU) ride
along under ``weak_1024`` / ``seq_10000`` / ``smplx_1024`` (measured in the same process after the timed
region, same K / W / repeats).  See DESIGN.md "Measurement" for `roofline` / `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

# algorithmic work per unit (SURVEY.md §8d / DESIGN.md)
# analytic forward + backward + Adam per frame and iteration.  SMPL: mixture 8 x 69 x 69 x 2 = 76 k + chain / Rodrigues 12 k +
# J(beta) 3 k + loss / Adam 2 k.  SMPL-X: mixture over 63 dimensions 64 k + 55-joint chain 27 k + J(shape) 13 k + loss / Adam 2 k.
FIT_FLOP_PER_FRAME_ITER = {"smpl": 0.11e6, "smplx": 0.106e6}
FP32_PEAK_TFLOPS = 157.3                # MI355X_MICROARCH.md: fp32 vector = fp32-input MFMA peak
HBM_PEAK_GBS = 8000.0
ROUND = "r04"
LBS_KERNELS = {"smpl": "k2b_pose_setup_kernel+k2b_lbs_stream_kernel", "smplx": "k2b_pose_setup_kernel+k2b_lbs_tile_kernel"}
PREWARM_S = 0.3        # seconds of untimed load before the warm-up steps (device clock ramp, see measure())
# HIP events around the fit launch, the forward and the exchange are recorded on every EVENT_EVERY-th step of the timed region
# (they are what `roofline.avg_launch_ms` averages: 5 of the default 20 steps of each of the nine blocks).  Four timing events per
# step cost ~10 us of a 0.57 ms step (same box, back to back: 0.5727 / 0.5746 ms with events on every step, 0.5587 / 0.5653 on
# every fourth) - instrumentation, not work of the path.  K2B_BENCH_EVENT_EVERY=1 records them on every step.
EVENT_EVERY = max(1, int(os.environ.get("K2B_BENCH_EVENT_EVERY", "4")))


def lbs_bytes_per_frame(model) -> int:
    """params in + joints out + vertices out (SMPL: 340 + 540 + 82 680 = 83 560 B)."""
    n = model.native
    p = 4 * (3 + 3 * (n.num_joints - 1) + n.num_betas + 3)
    return p + 12 * (n.num_joints + n.num_extra) + 12 * n.num_vertices


def lbs_flop_per_frame(model) -> float:
    """fp32-equivalent flops of the two LBS contractions + skinning (SMPL: 13.1 MFLOP)."""
    n = model.native
    V, J = n.num_vertices, n.num_joints
    return 2.0 * V * 3 * (9 * (J - 1) + n.num_betas) + 2.0 * V * 12 * J + 2.0 * V * 12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (a fresh process starts at a low device clock: see PREWARM_S; the timed block of K steps is repeated, see --repeats)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=9, help="timed blocks of K steps each; the median block is reported")
    ap.add_argument("--total-frames", type=int, default=4096,
                    help="frames of the ONE sequence sharded over all ranks (strong scaling; north-star: 4096)")
    ap.add_argument("--frames", type=int, default=None,
                    help="weak scaling instead: this many frames per GPU (BASELINE configs[1]: 1024)")
    ap.add_argument("--iters", type=int, default=100, help="Adam iterations per frame")
    ap.add_argument("--model", choices=("smpl", "smplx"), default="smpl")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-weak-line", action="store_true",
                    help="skip the secondary measurements (weak_1024, seq_10000, smplx_1024)")
    ap.add_argument("--cpu-runs", type=int, default=5, help="timed B=1 fits per thread setting of the CPU baseline")
    return ap.parse_args()


def build_problem(total_frames, start, stop, seed, device, model_kind="smpl"):
    """Synthetic model, prior and the frames [start, stop) of the `total_frames`-frame sequence `seed`."""
    from keypoints2body_amd import synthetic
    from keypoints2body_amd.core.fitters.world_space import guess_init_transl_from_root
    from keypoints2body_amd.models.body_model import BodyModel
    from keypoints2body_amd.models.smpl_data import SMPLData
    from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers

    if model_kind == "smplx":
        return build_problem_smplx(total_frames, start, stop, seed, device)
    model = BodyModel.synthetic(seed=0, device=device)
    gmm = synthetic.make_gmm(seed=0)
    prior = MaxMixturePrior(MixtureBuffers.from_mixture(gmm.means, gmm.covars, gmm.weights), device=device)
    poses = synthetic.make_poses(total_frames, seed=seed)
    n = stop - start
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a[start:stop]), dtype=torch.float32, device=device).contiguous()
    zeros = lambda c: torch.zeros((n, c), dtype=torch.float32, device=device)
    if n == 0:
        j3d = torch.zeros((0, 22, 3), dtype=torch.float32, device=device)
        transl0 = zeros(3)
    else:
        gt = model(global_orient=dev(poses.global_orient), body_pose=dev(poses.body_pose), betas=dev(poses.betas),
                   transl=dev(poses.transl), return_verts=False)
        j3d = gt.joints[:, :22].contiguous()
        transl0 = guess_init_transl_from_root(model, zeros(72), zeros(10), j3d, joints_category="AMASS")
    init = SMPLData(betas=zeros(10), global_orient=zeros(3), body_pose=zeros(69), transl=transl0.contiguous())
    return model, prior, j3d, init


def build_problem_smplx(total_frames, start, stop, seed, device):
    """BASELINE config 4: SMPL-X-shaped model (55 joints, V = 10475, 10 betas + 10 expression coefficients), all 55 kinematic
    joints observed (body, jaw, eyes, both hands), zero initialisation with a root-aligned translation."""
    from keypoints2body_amd import synthetic
    from keypoints2body_amd.models.body_model import BodyModel
    from keypoints2body_amd.models.smpl_data import SMPLXData
    from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers

    model = BodyModel.synthetic_x(seed=0, device=device)
    gmm = synthetic.make_gmm(seed=0)
    prior = MaxMixturePrior(MixtureBuffers.from_mixture(gmm.means, gmm.covars, gmm.weights), device=device)
    poses = synthetic.make_poses_x(total_frames, seed=seed)
    n = stop - start
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a[start:stop]), dtype=torch.float32, device=device).contiguous()
    zeros = lambda c: torch.zeros((n, c), dtype=torch.float32, device=device)
    fields = ("global_orient", "body_pose", "jaw_pose", "leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose", "betas",
              "expression", "transl")
    gt = model(**{k: dev(getattr(poses, k)) for k in fields}, return_verts=False)
    j3d = gt.joints[:, :55].contiguous()
    rest = model(global_orient=zeros(3), return_verts=False).joints
    transl0 = (j3d[:, 0] - rest[:, 0]).contiguous()
    init = SMPLXData(betas=zeros(10), global_orient=zeros(3), body_pose=zeros(63), transl=transl0, left_hand_pose=zeros(45),
                     right_hand_pose=zeros(45), expression=zeros(10), jaw_pose=zeros(3), leye_pose=zeros(3), reye_pose=zeros(3))
    return model, prior, j3d, init


def cpu_baseline(iters, runs):
    """Reference-style CPU loop (oracle port of world_space.py:248-256: full SMPL forward incl. all vertices every
    iteration, torch autograd, torch.optim.Adam) on this host's cores.  Per-frame B=1 fits exactly as
    `optimize_params_frame` does them: one warm-up fit, then the MEDIAN of `runs` timed fits, with all the cores
    this process may use and again with ONE thread; plus the best batched figure (B=32, median of 3)."""
    from keypoints2body_amd import synthetic
    from oracle.fit_torch import GMMPrior, fit_world_adam, guess_init_transl
    from oracle.smpl_torch import TorchSMPL

    # the GPU box gives one GPU's share of the host (16 CPUs); os.cpu_count() reports the whole machine
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    model = TorchSMPL(synthetic.make_body_model(0))
    g = synthetic.make_gmm(0)
    prior = GMMPrior(g.means, g.covars, g.weights)
    nb = 32
    poses = synthetic.make_poses(max(runs + 1, nb), seed=1000)
    t = lambda a: torch.tensor(a)
    with torch.no_grad():
        j3d = model(global_orient=t(poses.global_orient), body_pose=t(poses.body_pose), betas=t(poses.betas),
                    transl=t(poses.transl)).joints[:, :22].clone()
    B = j3d.shape[0]
    tr0 = guess_init_transl(model, torch.zeros(B, 72), torch.zeros(B, 10), j3d)
    z = lambda n, c: torch.zeros(n, c)

    def fit(sl):
        n = j3d[sl].shape[0]
        t0 = time.perf_counter()
        fit_world_adam(model, prior, z(n, 3), z(n, 69), z(n, 10), tr0[sl], j3d[sl], None, num_iters=iters)
        return time.perf_counter() - t0

    def loop(threads):
        torch.set_num_threads(threads)
        fit(slice(0, 1))                                     # warm-up
        return statistics.median(fit(slice(1 + i, 2 + i)) for i in range(runs))

    t_all = loop(cores)
    t_one = loop(1)
    torch.set_num_threads(cores)
    fit(slice(0, nb))
    t_batch = statistics.median(fit(slice(0, nb)) for _ in range(3))
    return {
        "value": round(1.0 / t_all, 3),
        "unit": "frames/s",
        "cores": cores,
        "kind": "port",
        "sample": (f"B=1 fits ({iters} Adam iters, full SMPL forward incl. 6890 vertices every iteration, torch autograd + "
                   f"torch.optim.Adam), 1 warm-up + median of {runs} timed fits: {t_all:.3f} s/frame on {cores} threads, "
                   f"{t_one:.3f} s/frame on 1 thread; one batched B={nb} call (median of 3): {nb / t_batch:.2f} frames/s"),
        "value_1thread": round(1.0 / t_one, 3),
        "batched_value": round(nb / t_batch, 3),
    }


def read_traffic(model_kind, frames):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/README.md); only known for the
    frame counts that were profiled."""
    for name in (f"traffic_{ROUND}.json", "traffic_r02.json", "traffic_r01.json"):
        tfile = REPO / "profiles" / name
        if not tfile.exists():
            continue
        try:
            tj = json.loads(tfile.read_text())
        except Exception:
            continue
        pre = "" if model_kind == "smpl" else model_kind + "_"
        fit, lbs = tj.get(f"{pre}fit_frames_{frames}"), tj.get(f"{pre}lbs_frames_{frames}")
        if fit is not None or lbs is not None:
            return fit, lbs, name
    return None, None, None


def read_sq(model_kind, frames):
    """Issue-side counters of the fit kernel at this size, derived from the committed SQ passes (profiles/<ROUND>_sq_fit_*.csv,
    tools/pmc_fit.sh): `valu_active_frac` = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of the waves' lifetime with a vector
    instruction executing), `mfma_busy_frac` = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x launch cycles at the 2.4 GHz peak clock)."""
    if model_kind != "smpl":
        return None
    f = REPO / "profiles" / f"{ROUND}_sq_fit.json"
    if not f.exists():
        return None
    try:
        return json.loads(f.read_text()).get(str(frames))
    except Exception:
        return None


def spawn_ranks(n):
    """``python bench.py --gpus N`` without a launcher: start N fresh rank processes (``torch.distributed.run``, one per GPU,
    rendezvous on 127.0.0.1) BEFORE this process has touched the GPU, relay their output (rank 0 prints the JSON line) and
    return the launcher's exit code (non-zero if any rank failed).  Nothing is re-exec'ed: the ranks are children."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    print(f"[bench] --gpus {n} without WORLD_SIZE: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        backend = os.environ.get("K2B_BENCH_BACKEND", "nccl")
        have = torch.cuda.device_count()                     # counting devices does not initialise the GPU
        if backend == "nccl" and have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible (RCCL needs one per rank)")
        raise SystemExit(spawn_ranks(args.gpus))
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

    from keypoints2body_amd import native
    from keypoints2body_amd.core.fitters.world_space import WorldSpaceFitter
    from keypoints2body_amd.parallel import fit_forward_exchange, shard_bounds

    ev = lambda: torch.cuda.Event(enable_timing=True)

    repeats = max(1, int(args.repeats))

    def measure(total, weak, steps, warmup, model_kind):
        """Time `steps` passes over this rank's share of `total` frames (weak: `total` = frames per GPU)."""
        if weak:
            T, start, stop, per = total * world, rank * total, (rank + 1) * total, total
        else:
            T = total
            start, stop = shard_bounds(T, world, rank)
            per = (T + world - 1) // world
        model, prior, j3d, init = build_problem(T, start, stop, 1000, device, model_kind)
        fitter = WorldSpaceFitter(model, step_size=1e-2, num_iters_first=args.iters, num_iters_followup=args.iters,
                                  use_lbfgs=False, joints_category="AMASS" if model_kind == "smpl" else "GENERIC",
                                  device=device, pose_prior=prior)
        cfg = fitter._config(0, 600.0, 5.0, False, False)
        K = j3d.shape[1]
        idx = list(range(K))
        if model_kind == "smplx":
            cfg.prior_pose_dims, cfg.num_betas_prior = 63, 10
            init = fitter.packed_init(init)
        fit_ev, lbs_ev, xch_ev = [], [], []

        def step(record=False, comm=True):
            """ONE pass of the hot path over this rank's block = ``parallel.fit_forward_exchange``, the very function the public
            ``optimize_params_sequence`` runs for independent frames: fit launch, parameter all-gather enqueued on RCCL's stream,
            final forward over the OWN block, joints all-gather, wait.  ``record``: HIP events around the fit launch, the forward
            and the exchange of THIS step (the per-kernel durations of the roofline objects)."""
            if not record:
                ex = fit_forward_exchange(lambda: fitter.fit_params(cfg, j3d, init, idx), fitter.final_forward, dist if comm else None, pad_to=per)
                return ex["local"], ex["joints"], ex["vertices"]
            e0, e1, e2, e3 = ev(), ev(), ev(), ev()

            def fit_fn():
                out = fitter.fit_params(cfg, j3d, init, idx)
                e1.record()
                return out

            def forward_fn(out):
                jv = fitter.final_forward(out)
                e2.record()
                return jv

            e0.record()
            ex = fit_forward_exchange(fit_fn, forward_fn, dist if comm else None, pad_to=per)
            e3.record()
            if record:
                fit_ev.append((e0, e1))
                lbs_ev.append((e1, e2))
                xch_ev.append((e2, e3))
            return ex["local"], ex["joints"], ex["vertices"]

        # clock ramp: a fresh process starts at a low device clock and needs ~0.1 s of load to reach the steady one (a
        # 1024-frame run with only W = 30 warm-up steps = 9 ms of work measured 0.391 ms/step, the same steps behind
        # 0.16 s of other work 0.313).  So the device is kept busy with this very step for PREWARM_S seconds first - set-up,
        # like building the problem, not part of the W warm-up steps or the K timed ones; `config.prewarm_s` states it.
        # (no collective in it: the number of pre-warm steps is time-based and differs between ranks)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < PREWARM_S:
            for _ in range(10):
                step(comm=False)
            torch.cuda.synchronize()
        for _ in range(warmup):
            step()
        # The timed region: EXACTLY `steps` steps between barrier + synchronize on both sides, maximum over the ranks - taken
        # `repeats` times back to back, and the MEDIAN block is the reported one (`repeats`, `block_ms` in the JSON line).  The
        # device shows transient slow phases of 20-40 ms in which every kernel takes about twice as long (clock / power state
        # transitions; seen on some boxes and not on others: the same 4096-frame fit launch 0.50 ms, then 0.94 ms for one chunk
        # of 50 launches, then 0.50 ms again - tools/dev_fit_sustained.py).  One window of any length either contains such a
        # phase or not (single windows of this workload read 3.3, 4.5 and 5.7 M frames/s within a minute on one box); the
        # median of nine short windows does not depend on it.
        blocks = []
        for rep in range(repeats):
            del fit_ev[:], lbs_ev[:], xch_ev[:]
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                out, joints, verts = step(record=i % EVENT_EVERY == 0)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            elapsed = time.perf_counter() - t0
            if dist is not None:
                tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                elapsed = float(tmax.item())
            mean_ms = lambda evs: float(np.mean([a.elapsed_time(b) for a, b in evs]))      # HIP events on the launch stream
            blocks.append((elapsed, mean_ms(fit_ev), mean_ms(lbs_ev), mean_ms(xch_ev)))
        order = sorted(range(repeats), key=lambda i: blocks[i][0])
        elapsed, fit_ms, lbs_ms, xch_ms = blocks[order[repeats // 2]]           # the median block (same block for all figures)
        n_local = stop - start
        jl = joints if joints.shape[0] == n_local else joints[rank * per: rank * per + n_local]   # gathered joints: own rows
        res = {
            "T": T, "frames_local": n_local, "elapsed": elapsed, "model": model, "kind": model_kind, "weak": weak,
            "fit_ms": fit_ms, "lbs_ms": lbs_ms, "exchange_ms": xch_ms,
            "block_ms": [round(1e3 * b[0], 3) for b in blocks],
            "err_cm": float((jl[:, :K] - j3d).norm(dim=-1).mean().item() * 100) if n_local else 0.0,
            "loss": float(out["loss"].mean()) if n_local else 0.0,
        }
        return res

    def fractions(r):
        """Roofline fractions of one measurement: fit kernel against the fp32 vector peak, the LBS launches against 8 TB/s."""
        F = r["frames_local"]
        flop_iter = FIT_FLOP_PER_FRAME_ITER[r["kind"]]
        lbs_bytes = lbs_bytes_per_frame(r["model"])
        fit_tflops = flop_iter * args.iters * F / (r["fit_ms"] * 1e-3) / 1e12
        lbs_gbs = lbs_bytes * F / (r["lbs_ms"] * 1e-3) / 1e9
        return flop_iter, lbs_bytes, fit_tflops, lbs_gbs

    def sub_line(r, workload):
        _, _, fit_tflops, lbs_gbs = fractions(r)
        d = {"workload": workload, "value": round(r["T"] * args.steps / r["elapsed"], 1), "unit": "frames/s",
             "scaling": "weak" if r["weak"] else "strong", "total_frames": r["T"], "frames_rank0": r["frames_local"],
             "ms_per_step": round(r["elapsed"] / args.steps * 1e3, 4),
             "fit_ms": round(r["fit_ms"], 4), "lbs_ms": round(r["lbs_ms"], 4),
             "fit_frac_fp32": round(fit_tflops / FP32_PEAK_TFLOPS, 4), "lbs_frac_hbm": round(lbs_gbs / HBM_PEAK_GBS, 4),
             "mean_joint_error_cm": round(r["err_cm"], 3)}
        if world > 1:
            d["exchange_ms"] = round(r["exchange_ms"], 4)
        return d

    weak = args.frames is not None
    main_res = measure(args.frames if weak else args.total_frames, weak, args.steps, args.warmup, args.model)
    extra = {}
    if not weak and not args.no_weak_line and args.model == "smpl":
        # the other BASELINE configs ride along in the same process (same steps / warm-up / repeats; each builds its own problem)
        extra["weak_1024"] = sub_line(measure(1024, True, args.steps, args.warmup, "smpl"),
                                      "BASELINE configs[1]: 1024 frames per GPU (weak scaling), same model and settings")
        extra["seq_10000"] = sub_line(measure(10000, False, args.steps, args.warmup, "smpl"),
                                      f"BASELINE configs[2]: 10 000-frame AMASS-style sequence, frames sharded over {world} GPU(s)")
        extra["smplx_1024"] = sub_line(measure(1024, True, args.steps, args.warmup, "smplx"),
                                       "BASELINE configs[3]: SMPL-X (55 joints, V=10475, betas | expression = 20), 1024 frames per GPU, "
                                       "all 55 kinematic joints observed, 100 Adam iters")

    if rank == 0:
        r = main_res
        model = r["model"]
        F = r["frames_local"]
        ms_per_step = r["elapsed"] / args.steps * 1e3
        flop_iter, lbs_bytes, fit_tflops, lbs_gbs = fractions(r)
        traffic, traffic_lbs, traffic_src = read_traffic(args.model, F)
        n = model.native
        shape = (f"{'SMPL' if args.model == 'smpl' else 'SMPL-X'}-shaped model (V={n.num_vertices}, J={n.num_joints}, "
                 f"{n.num_betas} betas)")
        targets = "22-joint AMASS" if args.model == "smpl" else "55-joint SMPL-X (body + jaw + eyes + hands)"
        if weak:
            workload = (f"{args.frames} synthetic {targets} frames per GPU, {shape}, {args.iters} Adam iters, "
                        "world mode, final joints+vertices produced")
        else:
            workload = (f"{r['T']}-frame sequence of synthetic {targets} frames sharded over {world} GPU(s) "
                        f"({F} frames on rank 0), {shape}, {args.iters} Adam iters, world mode, final joints+vertices produced")
        line = {
            "metric": "SMPL frames fitted/sec (100 Adam iters, 22-joint AMASS)" if args.model == "smpl"
                      else "SMPL-X frames fitted/sec (100 Adam iters, 55 kinematic joints)",
            "value": round(r["T"] * args.steps / r["elapsed"], 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "repeats": repeats,                 # timed blocks of `steps` steps; value / ms_per_step are the median block's
            "event_every": EVENT_EVERY,         # steps between HIP-event samples of the per-kernel durations (roofline.avg_launch_ms)
            "block_ms": r["block_ms"],
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "total_frames": r["T"], "frames_rank0": F, "adam_iters": args.iters,
                "parallelism": f"frames sharded x{world}",
                "prewarm_s": PREWARM_S,          # untimed load before the W warm-up steps (device clock ramp)
                "step": "parallel.fit_forward_exchange (the function optimize_params_sequence runs for independent frames): "
                        "fit own block, parameter all-gather under the final forward of the own block, joints all-gather",
            },
            # dominant kernel: the fused fit.  It never touches HBM inside its loop; its bound is fp32 vector-ALU issue
            # (DESIGN §4.1), so the peak is the fp32 VALU peak (= the fp32-input MFMA peak), not an f16 matrix peak.
            "roofline": {
                "kernel": "k2b_fit_world_kernel" if args.model == "smpl" else "k2b_fit_tree_kernel", "bound": "valu", "achieved": round(fit_tflops, 3),
                "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fit_tflops / FP32_PEAK_TFLOPS, 4),
                "traffic": traffic, "avg_launch_ms": round(r["fit_ms"], 4),
                "note": f"algorithmic fp32 flops ({flop_iter / 1e6:.3f} MFLOP per frame-iteration) against the fp32 vector peak",
            },
            "roofline_lbs": {
                "kernel": LBS_KERNELS[args.model], "bound": "hbm",
                "achieved": round(lbs_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(lbs_gbs / HBM_PEAK_GBS, 4), "traffic": traffic_lbs, "avg_launch_ms": round(r["lbs_ms"], 4),
                "bytes_per_frame": lbs_bytes,
                "achieved_tflops": round(lbs_flop_per_frame(model) * F / (r["lbs_ms"] * 1e-3) / 1e12, 2),
            },
            "quality": {"mean_joint_error_cm": round(r["err_cm"], 3), "mean_final_loss": round(r["loss"], 2)},
        }
        if world > 1:
            line["exchange_ms"] = round(r["exchange_ms"], 4)    # what the step waits for the collectives behind the forward
        if traffic_src:
            line["roofline"]["traffic_source"] = line["roofline_lbs"]["traffic_source"] = "profiles/" + traffic_src
        sq = read_sq(args.model, F)
        if sq:
            line["roofline"].update({k: sq[k] for k in ("valu_active_frac", "mfma_busy_frac", "wait_any_frac", "sq_source") if k in sq})
        line.update(extra)
        if world == 1 and not args.no_cpu_baseline and args.model == "smpl":
            line["cpu_baseline"] = cpu_baseline(args.iters, args.cpu_runs)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
