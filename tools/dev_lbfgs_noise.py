"""Spread of the LBFGS-mode final loss under rounding-level perturbations of the inputs (HIP path)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from tests import helpers as H
import keypoints2body_amd as k2b
from keypoints2body_amd.core.fitters.world_space import WorldSpaceFitter
from keypoints2body_amd.models.body_model import BodyModel
from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers
g = H.gmm_fixture()
prior = MaxMixturePrior(MixtureBuffers(g["ref_means"], g["ref_precisions"], g["ref_nll_weights"].reshape(-1)))
model = BodyModel.synthetic(0)
for case in ('first', 'followup', 'frozen'):
    d = dict(np.load(H.GOLDEN / f'lbfgs_world_{case}.npz'))
    it = int(d['max_iter'])
    fitter = WorldSpaceFitter(model, step_size=1e-2, num_iters_first=it, num_iters_followup=it, use_lbfgs=True, joints_category='AMASS', pose_prior=prior)
    torch.manual_seed(0)
    for i in range(d['j3d'].shape[0]):
        t = lambda k: torch.tensor(d[k][i:i + 1])
        losses = []
        for trial in range(8):
            eps = 0.0 if trial == 0 else 2e-6
            res = fitter.fit_frame(k2b.SMPLData(betas=t('init_betas'), global_orient=t('init_global_orient') * (1 + eps * torch.randn(1, 3)),
                                                body_pose=t('init_body_pose') * (1 + eps * torch.randn(1, 69)), transl=t('init_transl') * (1 + eps * torch.randn(1, 3))),
                                   t('j3d'), conf_3d=torch.tensor(d['conf']), seq_ind=int(d['seq_ind']), freeze_betas=bool(int(d['freeze_betas'])))
            losses.append(float(res.loss))
        print(case, i, 'ref', float(d['out_loss'][i]), 'hip runs', np.round(losses, 1))
