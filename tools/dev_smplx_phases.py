"""Development: SMPL-X tree-kernel fit at 1024 frames with and without the mixture prior (how much of an iteration is the
component phase between the two barriers)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from tests import helpers as H
from keypoints2body_amd import native, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m, pr = H.native_model_x(), H.native_prior()
p = synthetic.make_poses_x(B, seed=9)
POSE = ("body_pose", "jaw_pose", "leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose")
pose = np.concatenate([getattr(p, k) for k in POSE], axis=1)
shape = np.concatenate([p.betas, p.expression], axis=1)
j, _ = m.lbs(H.cuda(p.global_orient), H.cuda(pose), H.cuda(shape), H.cuda(p.transl), want_vertices=False)
j3d = j[:, :55].contiguous()
z = lambda c: torch.zeros(B, c, device="cuda")
for name, w in (("with the mixture prior", None), ("pose_prior_weight = 0", 0.0)):
    cfg = native.default_fit_config(); cfg.num_iters = 100; cfg.prior_pose_dims, cfg.num_betas_prior = 63, 10
    if w is not None: cfg.pose_prior_weight = w
    run = lambda: native.fit_world(m, pr, cfg, list(range(55)), j3d, None, z(3), z(162), z(20), j3d[:, 0].contiguous())
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{B} SMPL-X frames, {name}: fit {e0.elapsed_time(e1) / 50:.4f} ms")
