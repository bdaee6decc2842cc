"""Wall time per frame of a camera-mode sequence (warm-start chain = the reference's default, and independent frames), Adam and
device L-BFGS.  usage: python3 tools/dev_camera_seq.py [frames] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from tests import helpers as H
import keypoints2body_amd as k2b
from keypoints2body_amd import synthetic
from keypoints2body_amd.core.config import FrameOptimizeConfig, SequenceOptimizeConfig
from keypoints2body_amd.models.body_model import BodyModel
from keypoints2body_amd.prior import MaxMixturePrior, MixtureBuffers

T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
g = H.gmm_fixture()
prior = MaxMixturePrior(MixtureBuffers(g["ref_means"], g["ref_precisions"], g["ref_nll_weights"].reshape(-1)))
model = BodyModel.synthetic(0)
j = synthetic.make_sequence_targets(T, seed=5) if hasattr(synthetic, "make_sequence_targets") else None
if j is None:
    p = synthetic.make_poses(T, seed=31)
    with torch.no_grad():
        j = H.oracle_model()(global_orient=torch.tensor(p.global_orient), body_pose=torch.tensor(p.body_pose),
                             betas=torch.tensor(p.betas), transl=torch.tensor(p.transl)).joints[:, :22].numpy()
seq = np.concatenate([np.asarray(j, np.float32), np.ones((T, 22, 1), np.float32)], axis=2)
mean = (torch.zeros(1, 72), torch.zeros(1, 10))
for warm in (True, False):
    for use_lbfgs in (False, True):
        cfg = SequenceOptimizeConfig(frame=FrameOptimizeConfig(use_lbfgs=use_lbfgs, coordinate_mode="camera", num_iters=iters,
                                                               joints_category="AMASS"),
                                     use_previous_frame_init=warm, use_shape_optimization=False, fix_foot=False)
        run = lambda: k2b.optimize_params_sequence(seq, joint_layout="AMASS", model=model, config=cfg, pose_prior=prior, mean_params=mean)
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter(); r = run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"camera sequence T={T} iters={iters} warm_start={warm} lbfgs={use_lbfgs}: {dt * 1e3:.2f} ms = {dt / T * 1e6:.1f} us/frame, "
              f"loss last {float(r[-1].loss):.4f}")

if len(sys.argv) > 3:      # host profile of the warm-start Adam chain
    import cProfile, pstats, io
    cfg = SequenceOptimizeConfig(frame=FrameOptimizeConfig(use_lbfgs=False, coordinate_mode="camera", num_iters=iters, joints_category="AMASS"),
                                 use_previous_frame_init=True, use_shape_optimization=False, fix_foot=False)
    run = lambda: k2b.optimize_params_sequence(seq, joint_layout="AMASS", model=model, config=cfg, pose_prior=prior, mean_params=mean)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e6 * (t1 - t0) / T:.1f} us/frame, with sync {1e6 * (t2 - t0) / T:.1f} us/frame")
    pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(30); print(st.getvalue()[:7000])
