"""Where two launch shapes of the fit kernel differ after ONE iteration (loss, gradient, parameters): usage dev_shape_diff.py A B [frames]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from tests import helpers as H
from keypoints2body_amd import native, synthetic
A, Bs = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 37
m, pr = H.native_model(), H.native_prior()
p = synthetic.make_poses(B, seed=11)
go, bp, be, tr = map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl))
j, _ = m.lbs(go, bp, be, tr, want_vertices=False)
j3d = (j[:, :22] + 0.01).contiguous()
res = {}
for it in (1, 3):
    for shape in (A, Bs):
        cfg = native.default_fit_config(); cfg.num_iters = it; cfg.debug_launch_shape = shape
        res[shape] = native.fit_world(m, pr, cfg, list(range(22)), j3d, None, go * 0.9, bp * 0.9, be * 0.5, tr, want_grad=True)
    for k in ("loss", "grad", "global_orient", "body_pose", "betas", "transl"):
        a, b = res[A][k].cpu().numpy(), res[Bs][k].cpu().numpy()
        d = np.abs(a - b)
        where = np.argwhere(d > 0)
        print(f"iters {it} {k}: max diff {d.max():.3e}; differing entries {len(where)}; first {where[:6].tolist()}")
    g = np.abs(res[A]["grad"].cpu().numpy() - res[Bs]["grad"].cpu().numpy())
    print("  grad columns that differ:", sorted(set(np.argwhere(g > 0)[:, 1].tolist()))[:40], " frames:", sorted(set(np.argwhere(g > 0)[:, 0].tolist()))[:40])
