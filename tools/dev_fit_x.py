"""Time the SMPL-X tree-kernel fit at one batch size: dev_fit_x.py FRAMES [shape 0|1|2] [launches] [lib variant]
(shape: k2b_fit_config.debug_launch_shape - 0 by batch size, 1 plain, 2 component waves)"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tests import helpers as H
from keypoints2body_amd import native, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shape = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
if len(sys.argv) > 4:
    native._LIB_PATH = Path(__file__).resolve().parent / f"libk2b_{sys.argv[4]}.so"
m, pr = H.native_model_x(), H.native_prior()
p = synthetic.make_poses_x(B, seed=1)
pose = np.concatenate([p.body_pose, p.jaw_pose, p.leye_pose, p.reye_pose, p.left_hand_pose, p.right_hand_pose], axis=1)
go, bp, be, tr = map(H.cuda, (p.global_orient, pose, np.concatenate([p.betas, p.expression], axis=1), p.transl))
j, _ = m.lbs(go, bp, be, tr, want_vertices=False)
j3d = j[:, :55].contiguous()
z = lambda *s: torch.zeros(*s, device="cuda")
j0, _ = m.lbs(z(B, 3), z(B, 162), z(B, 20), None, want_vertices=False)
tr0 = (j3d[:, 0] - j0[:, 0]).contiguous()
cfg = native.default_fit_config(); cfg.num_iters = 100; cfg.prior_pose_dims, cfg.num_betas_prior = 63, 10
cfg.debug_launch_shape = shape
run = lambda: native.fit_world(m, pr, cfg, list(range(55)), j3d, None, z(B, 3), z(B, 162), z(B, 20), tr0)
o = run(); torch.cuda.synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    for _ in range(5): run()
    torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(n): o = run()
ev[1].record(); torch.cuda.synchronize()
print(f"smplx {sys.argv[4] if len(sys.argv) > 4 else 'head'} B={B} shape={shape}: fit {ev[0].elapsed_time(ev[1]) / n:.4f} ms  loss mean {o['loss'].mean().item():.2f}")
