"""The reference's DEFAULT sequence configuration (use_lbfgs=True, use_previous_frame_init=True: core/config.py:29,57) on a
195-frame synthetic motion through the public optimize_params_sequence: the one-call device chain (k2b_fit_sequence_lbfgs)
against one device-driven fit per frame and against the round-3 host driver (torch.optim.LBFGS per closure call, 20 frames)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import keypoints2body_amd as k2b
from keypoints2body_amd import synthetic
from keypoints2body_amd.core.config import FrameOptimizeConfig, SequenceOptimizeConfig
from keypoints2body_amd.core.fitters.world_space import WorldSpaceFitter
from keypoints2body_amd.models.body_model import BodyModel
from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers
from tests import helpers as H
g = H.gmm_fixture()
prior = MaxMixturePrior(MixtureBuffers(g["ref_means"], g["ref_precisions"], g["ref_nll_weights"].reshape(-1)))
model = BodyModel.synthetic(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 195
p = synthetic.make_poses(1, seed=17)
walk = np.cumsum(0.02 * np.random.default_rng(2).standard_normal((T, 69)), axis=0).astype(np.float32)
with torch.no_grad():
    j = H.oracle_model()(global_orient=torch.tensor(np.repeat(p.global_orient, T, 0)), body_pose=torch.tensor(p.body_pose + walk),
                         betas=torch.tensor(np.repeat(p.betas, T, 0)), transl=torch.tensor(np.repeat(p.transl, T, 0))).joints[:, :22]
seq = np.concatenate([j.numpy(), np.ones((T, 22, 1), np.float32)], axis=2)
cfg = SequenceOptimizeConfig(frame=FrameOptimizeConfig(joints_category="AMASS"), use_shape_optimization=False)   # defaults: L-BFGS 30 / 10, warm start
kw = dict(joint_layout="AMASS", model=model, config=cfg, pose_prior=prior, mean_params=(torch.zeros(1, 72), torch.zeros(1, 10)))
def run(n=T):
    t0 = time.perf_counter()
    r = k2b.optimize_params_sequence(seq[:n], **kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, r
run(8)
t_chain, r1 = run()
orig = WorldSpaceFitter.chain_supported
WorldSpaceFitter.chain_supported = lambda self, idx=None: False
run(8)
t_loop, r2 = run()
same = all(torch.equal(a.params.body_pose, b.params.body_pose) for a, b in zip(r1, r2))
WorldSpaceFitter.lbfgs_driver = "torch"
run(4)
t_host, _ = run(20)
print(f"{T} frames, reference default configuration: one-call device chain {1e3 * t_chain:.1f} ms ({1e3 * t_chain / T:.3f} ms/frame); "
      f"device fit per frame {1e3 * t_loop:.1f} ms ({1e3 * t_loop / T:.3f} ms/frame), same results {same}; "
      f"round-3 host driver {1e3 * t_host / 20:.2f} ms/frame (20 frames)")
