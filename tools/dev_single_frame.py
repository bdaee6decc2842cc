"""Development: wall time of ONE optimize_params_frame call (BASELINE configs[0]: a single 22-joint AMASS frame) on the GPU,
Adam branch (one fused launch) and L-BFGS branch (the reference default: host optimiser over evaluate-only launches)."""
import os, sys, time, statistics
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import keypoints2body_amd as k2b
from keypoints2body_amd.core.config import FrameOptimizeConfig
from keypoints2body_amd.models.body_model import BodyModel
from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers
from tests import helpers as H
g = H.gmm_fixture()
prior = MaxMixturePrior(MixtureBuffers(g["ref_means"], g["ref_precisions"], g["ref_nll_weights"].reshape(-1)))
model = BodyModel.synthetic(0)
d = H.load_case("amass_noisy_conf")
pose = torch.tensor(np.concatenate([d["init_global_orient"][:1], d["init_body_pose"][:1]], axis=1))
mean = (pose, torch.tensor(d["init_betas"][:1]))
joints = np.concatenate([d["j3d"][0], d["conf"][:, None]], axis=1)
for name, cfg in (("Adam, 100 iterations", dict(use_lbfgs=False, num_iters_first=100)),
                  ("L-BFGS, max_iter 30 (reference default)", dict(use_lbfgs=True, num_iters_first=30))):
    ts = []
    for i in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = k2b.optimize_params_frame(joints, joint_layout="AMASS", model=model, config=FrameOptimizeConfig(**cfg), pose_prior=prior,
                                        mean_params=mean)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{name}: median {1e3 * statistics.median(ts[2:]):.2f} ms per call (min {1e3 * min(ts[2:]):.2f}), final loss {float(res.loss):.1f}")
