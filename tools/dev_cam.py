import sys, numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tests import helpers as H
from keypoints2body_amd import native
from keypoints2body_amd.core.fitters.camera_space import CameraSpaceFitter
from keypoints2body_amd.models.body_model import BodyModel
from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers
from keypoints2body_amd.models.smpl_data import SMPLData
from oracle.fit_torch import fit_camera_adam_one
g = H.gmm_fixture()
prior = MaxMixturePrior(MixtureBuffers(g["ref_means"], g["ref_precisions"], g["ref_nll_weights"].reshape(-1)))
model = BodyModel.synthetic(0)
d = dict(np.load(H.GOLDEN / "camera_fit_short.npz"))
i = 0
t = lambda k: torch.tensor(d[k][i:i+1])
o = fit_camera_adam_one(H.oracle_model(), H.oracle_prior(), t("init_global_orient"), t("init_body_pose"), t("init_betas"), t("j3d"), None, num_iters=5)
print("oracle vs golden bp", (o.body_pose - t("out_body_pose")).abs().max().item())
f = CameraSpaceFitter(model, num_iters=5, use_lbfgs=False, joints_category="AMASS", pose_prior=prior)
# replicate stages by hand
go, bp, be = H.cuda(d["init_global_orient"][i:i+1]), H.cuda(d["init_body_pose"][i:i+1]), H.cuda(d["init_betas"][i:i+1])
j3d = H.cuda(d["j3d"][i:i+1])
res = f.fit_frame(SMPLData(betas=be, global_orient=go, body_pose=bp), j3d, freeze_betas=False)
print("init cam_t oracle", o.init_cam_t.numpy())
print("stage1 oracle go", o.stage1[0].numpy(), "t", o.stage1[1].numpy())
# run stage 1 only on HIP
from keypoints2body_amd.core.fitters import camera_space as cs
joints0 = model(global_orient=go, body_pose=bp, betas=be, return_verts=False).joints
t0 = cs.guess_init_3d(joints0, j3d, "AMASS").contiguous()
print("init cam_t hip", t0.cpu().numpy())
cfg = native.default_fit_config(); cfg.num_iters=5; cfg.sigma=1e8; cfg.joint_loss_weight=1.0
cfg.pose_prior_weight=cfg.angle_prior_weight=cfg.shape_prior_weight=cfg.pose_preserve_weight=0.0
cfg.optimize_mask=9; cfg.transl_prior_weight=200.0
s1 = native.fit_world(model.native, prior.native, cfg, cs._TORSO_IDX, j3d[:, cs._TORSO_IDX].contiguous(), None, go, bp, be, t0, transl_prior_target=t0, want_grad=True)
print("stage1 hip go", s1["global_orient"].cpu().numpy(), "t", s1["transl"].cpu().numpy())
print("stage1 bp unchanged", (s1["body_pose"].cpu()-bp.cpu()).abs().max().item(), "betas", (s1["betas"].cpu()-be.cpu()).abs().max().item())
diff = (res.params.body_pose.cpu() - t("out_body_pose")).abs()[0]
print("final bp diff top idx", torch.topk(diff, 8))
print("final go", res.params.global_orient.cpu().numpy(), d["out_global_orient"][i], "t", res.params.transl.cpu().numpy(), d["out_transl"][i])
