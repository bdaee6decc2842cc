"""Default-optimiser (L-BFGS) sequence of independent frames: the lock-step batch against one torch.optim.LBFGS per frame.
    python tools/dev_lbfgs_batch.py [FRAMES=256]"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import keypoints2body_amd as k2b
from keypoints2body_amd import synthetic
from keypoints2body_amd.models.body_model import BodyModel
from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers
from keypoints2body_amd.core.fitters.world_space import WorldSpaceFitter

T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = BodyModel.synthetic(0)
g = synthetic.make_gmm(0)
prior = MaxMixturePrior(MixtureBuffers.from_mixture(g.means, g.covars, g.weights))
poses = synthetic.make_poses(T, seed=3)
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=model.device)
j3d = model(global_orient=dev(poses.global_orient), body_pose=dev(poses.body_pose), betas=dev(poses.betas), transl=dev(poses.transl),
            return_verts=False).joints[:, :22].cpu().numpy()
cfg = {"use_shape_optimization": False, "use_previous_frame_init": False}      # frame defaults: use_lbfgs=True, 30 / 10 iterations
mean = (torch.zeros(1, 72), torch.zeros(1, 10))
run = lambda: k2b.optimize_params_sequence(j3d, body_model="smpl", joint_layout="AMASS", model=model, config=cfg, pose_prior=prior, mean_params=mean)
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); res = run(); torch.cuda.synchronize(); t_batch = time.perf_counter() - t0
err = np.mean([float((r.joints[:, :22].cpu() - torch.tensor(j3d[i:i + 1])).norm(dim=-1).mean()) for i, r in enumerate(res)])
print(f"lock-step: {T} frames {t_batch * 1e3:.1f} ms ({T / t_batch:.0f} frames/s), mean joint error {err * 100:.2f} cm")
# one optimiser per frame (the round-2 path): force B = 1 calls
fitter = WorldSpaceFitter(model, step_size=1e-2, num_iters_first=30, num_iters_followup=10, use_lbfgs=True, joints_category="AMASS", pose_prior=prior)
init = res[0].params
n = min(T, 64)
torch.cuda.synchronize(); t0 = time.perf_counter()
errs = []
for i in range(1, n):
    r = fitter.fit_frame(k2b.SMPLData(betas=torch.zeros(1, 10), global_orient=torch.zeros(1, 3), body_pose=torch.zeros(1, 69),
                                      transl=torch.tensor(j3d[0:1, 0])), torch.tensor(j3d[i:i + 1]), conf_3d=torch.ones(22), seq_ind=i)
    errs.append(float((r.joints[:, :22].cpu() - torch.tensor(j3d[i:i + 1])).norm(dim=-1).mean()))
torch.cuda.synchronize(); t_one = (time.perf_counter() - t0) / (n - 1)
print(f"per frame: {t_one * 1e3:.2f} ms per frame -> {T} frames {t_one * T * 1e3:.0f} ms; speed-up {t_one * T / t_batch:.1f} x; mean joint error {np.mean(errs) * 100:.2f} cm")
