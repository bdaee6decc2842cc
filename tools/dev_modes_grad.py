"""Compare loss/gradient of the fit-kernel shapes against the split one (evaluate-only launches)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from tests import helpers as H
from keypoints2body_amd import native, synthetic
m, pr = H.native_model(), H.native_prior()
for B in (1, 2, 3, 5, 17, 40):
    p = synthetic.make_poses(B, seed=3)
    go, bp, be, tr = map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl))
    j, _ = m.lbs(go, bp, be, tr, want_vertices=False)
    j3d = (j[:, :22] + 0.01).contiguous()
    cfg = native.default_fit_config(); cfg.num_iters = 1; cfg.step_size = 0.0
    res = {}
    for mode in ('split', 'split_paired', 'paired'):
        cfg.debug_launch_shape = H.LAUNCH_SHAPES[mode]
        o = native.fit_world(m, pr, cfg, list(range(22)), j3d, None, go * 0.9, bp * 0.9, be * 0.5, tr, want_grad=True)
        res[mode] = (o['loss'].cpu().numpy(), o['grad'].cpu().numpy())
    l0, g0 = res['split']
    for mode in ('split_paired', 'paired'):
        l, g = res[mode]
        d = np.abs(g - g0)
        sc = np.abs(g0).max()
        print(f'B={B} {mode}: loss rel {np.abs(l - l0).max() / np.abs(l0).max():.1e}  grad go {d[:, :3].max() / sc:.1e} bp {d[:, 3:72].max() / sc:.1e} betas {d[:, 72:82].max() / sc:.1e} transl {d[:, 82:].max() / sc:.1e}  worst frame {d.max(axis=1).argmax()} worst col {d.max(axis=0).argmax()}')
