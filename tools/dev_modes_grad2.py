import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from tests import helpers as H
from keypoints2body_amd import native, synthetic
np.set_printoptions(linewidth=200, precision=3, suppress=False)
m, pr = H.native_model(), H.native_prior()
B = 2
p = synthetic.make_poses(B, seed=3)
go, bp, be, tr = map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl))
j, _ = m.lbs(go, bp, be, tr, want_vertices=False)
j3d = (j[:, :22] + 0.01).contiguous()
cfg = native.default_fit_config(); cfg.num_iters = 1; cfg.step_size = 0.0
cfg.joint_loss_weight = 0.0; cfg.angle_prior_weight = 0.0; cfg.shape_prior_weight = 0.0
res = {}
for mode in ('unified', 'paired'):
    os.environ['K2B_FIT_MODE'] = mode
    o = native.fit_world(m, pr, cfg, list(range(22)), j3d, None, go * 0.9, bp * 0.9, be * 0.5, tr, want_grad=True)
    res[mode] = (o['loss'].cpu().numpy(), o['grad'].cpu().numpy())
print('loss', res['unified'][0], res['paired'][0])
for fidx in range(B):
    gu, gp = res['unified'][1][fidx, 3:72], res['paired'][1][fidx, 3:72]
    print('frame', fidx, 'unified bp grad', gu)
    print('frame', fidx, 'paired  bp grad', gp)
    print('ratio', gp / gu)
