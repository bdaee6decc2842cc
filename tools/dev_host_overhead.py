"""Development: host time to ENQUEUE one bench step (fit + final forward) vs the GPU time it takes: python tools/dev_host_overhead.py [frames]"""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from keypoints2body_amd.core.fitters.world_space import WorldSpaceFitter
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
model, prior, j3d, init = bench.build_problem(T, 0, T, 1000, dev, "smpl")
fitter = WorldSpaceFitter(model, step_size=1e-2, num_iters_first=100, num_iters_followup=100, use_lbfgs=False,
                          joints_category="AMASS", device=dev, pose_prior=prior)
cfg = fitter._config(0, 600.0, 5.0, False, False)
K = j3d.shape[1]
for _ in range(5):
    out = fitter.fit_params(cfg, j3d, init, list(range(K))); fitter.final_forward(out)
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n):
    out = fitter.fit_params(cfg, j3d, init, list(range(K)))
    fitter.final_forward(out)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{T} frames: host enqueue {1e3 * (t1 - t0) / n:.3f} ms per step, GPU-bound wall {1e3 * (t2 - t0) / n:.3f} ms per step")
for nev in (0, 1, 2, 3):
    evs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(nev)]
        if nev >= 1: e[0].record()
        out = fitter.fit_params(cfg, j3d, init, list(range(K)))
        if nev >= 2: e[1].record()
        fitter.final_forward(out)
        if nev >= 3: e[2].record()
        evs.append(e)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{T} frames, {nev} event records per step: wall {1e3 * (t2 - t0) / n:.3f} ms per step")
