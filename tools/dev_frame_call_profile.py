"""Development: where the host time of one optimize_params_frame call goes (cProfile over 300 calls, Adam and L-BFGS)."""
import os, sys, cProfile, pstats, io, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
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
cfg = FrameOptimizeConfig(use_lbfgs=True, num_iters_first=30)
call = lambda: k2b.optimize_params_frame(joints, joint_layout="AMASS", model=model, config=cfg, pose_prior=prior, mean_params=mean)
for _ in range(20): call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): call()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"300 calls: host {1e3 * (t1 - t0) / 300:.3f} ms per call enqueued, {1e3 * (t2 - t0) / 300:.3f} ms per call with the final sync")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): call()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
