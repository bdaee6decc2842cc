"""Development: warm-start chain (sequence mode) as one launch vs one launch per frame.  Usage: python tools/dev_chain_timing.py [T] [S]"""
import sys, time, copy
import torch
sys.path.insert(0, ".")
from tests import helpers as H
from tests.test_gpu_chain import problem, stepwise
from keypoints2body_amd import native

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
j3d, conf, go, bp, be, tr = problem(S, T, seed=1)
cfg = native.default_fit_config()
cfg.num_iters, cfg.pose_preserve_weight = 30, 5.0
idx = list(range(22))
for name, fn in (("one launch", lambda: native.fit_sequence(H.native_model(), H.native_prior(), cfg, 10, idx, j3d, conf, go, bp, be, tr)),
                 ("per frame ", lambda: stepwise(cfg, 10, idx, j3d, conf, go, bp, be, tr))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: S={S} T={T}  {dt * 1e3:9.2f} ms  = {dt / T * 1e6:8.2f} us / frame step   ({S * T / dt:,.0f} frames/s)")
