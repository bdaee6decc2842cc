"""Dev: time the LBS forward with a given library variant: dev_lbs_time.py <libname|-> <frames> [smpl|smplx]"""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from keypoints2body_amd import native
if sys.argv[1] != "-":
    native._LIB_PATH = Path(__file__).resolve().parent / f"libk2b_{sys.argv[1]}.so"
from tests import helpers as H
from keypoints2body_amd import synthetic
B = int(sys.argv[2])
KIND = sys.argv[3] if len(sys.argv) > 3 else "smpl"
if KIND == "smplx":
    import numpy as np
    m = H.native_model_x()
    p = synthetic.make_poses_x(B, seed=1)
    pose = np.concatenate([getattr(p, k) for k in ("body_pose", "jaw_pose", "leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose")], axis=1)
    args = list(map(H.cuda, (p.global_orient, pose, np.concatenate([p.betas, p.expression], axis=1), p.transl)))
else:
    m = H.native_model()
    p = synthetic.make_poses(B, seed=1)
    args = list(map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl)))
for _ in range(3): m.lbs(*args)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): m.lbs(*args)
e1.record(); torch.cuda.synchronize()
print(sys.argv[1], KIND, B, "frames: lbs", round(e0.elapsed_time(e1) / 50, 4), "ms")
