"""Dev: run-time ablation of the fit kernel (skip the GMM via pose_prior_weight = 0)."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tests import helpers as H
from keypoints2body_amd import native, synthetic
m, pr = H.native_model(), H.native_prior()
for B in (1024, 2048, 4096):
    p = synthetic.make_poses(B, seed=1)
    j, _ = m.lbs(*map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl)), want_vertices=False)
    j3d = j[:, :22].contiguous()
    z = lambda *s: torch.zeros(*s, device="cuda")
    for name, wpp in (("full", None), ("no GMM", 0.0)):
        cfg = native.default_fit_config(); cfg.num_iters = 100
        if wpp is not None: cfg.pose_prior_weight = wpp
        run = lambda: native.fit_world(m, pr, cfg, list(range(22)), j3d, None, z(B, 3), z(B, 69), z(B, 10), j3d[:, 0].contiguous())
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} {name}: {e0.elapsed_time(e1) / 5:.3f} ms")
