"""Development: does the fit launch slow down under sustained load?  Chunks of 50 launches (4096 frames), alone and alternating
with the LBS forward, timed with HIP events per chunk."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from keypoints2body_amd.core.fitters.world_space import WorldSpaceFitter
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
SHAPE = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # k2b_fit_config.debug_launch_shape
dev = torch.device("cuda:0")
model, prior, j3d, init = bench.build_problem(T, 0, T, 1000, dev, "smpl")
fitter = WorldSpaceFitter(model, step_size=1e-2, num_iters_first=100, num_iters_followup=100, use_lbfgs=False,
                          joints_category="AMASS", device=dev, pose_prior=prior)
cfg = fitter._config(0, 600.0, 5.0, False, False)
cfg.debug_launch_shape = SHAPE
K = j3d.shape[1]
ev = lambda: torch.cuda.Event(enable_timing=True)
for mode in ("fit only", "fit + LBS"):
    out = fitter.fit_params(cfg, j3d, init, list(range(K))); torch.cuda.synchronize()
    rows = []
    t0 = time.perf_counter()
    for chunk in range(8):
        fit_ms = 0.0
        pairs = []
        for _ in range(50):
            a, b = ev(), ev()
            a.record(); out = fitter.fit_params(cfg, j3d, init, list(range(K))); b.record()
            if mode != "fit only": fitter.final_forward(out)
            pairs.append((a, b))
        torch.cuda.synchronize()
        rows.append((time.perf_counter() - t0, sum(a.elapsed_time(b) for a, b in pairs) / 50))
    print(f"shape {SHAPE}", mode, " ".join(f"[{t:.2f}s {ms:.3f}]" for t, ms in rows))
