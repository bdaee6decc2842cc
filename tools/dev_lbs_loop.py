"""Development: LBS forward timed on a steady clock with a given library variant: dev_lbs_loop.py <libname|-> <frames> [x]
(third argument x: the SMPL-X-shaped model, 55 joints, V = 10475)"""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from keypoints2body_amd import native
if sys.argv[1] != "-":
    native._LIB_PATH = Path(__file__).resolve().parent / f"libk2b_{sys.argv[1]}.so"
from tests import helpers as H
from keypoints2body_amd import synthetic
B = int(sys.argv[2])
XMODEL = len(sys.argv) > 3 and sys.argv[3] == "x"
if XMODEL:
    import numpy as np
    m = H.native_model_x()
    p = synthetic.make_poses_x(B, seed=1)
    pose = np.concatenate([p.body_pose, p.jaw_pose, p.leye_pose, p.reye_pose, p.left_hand_pose, p.right_hand_pose], axis=1)
    args = list(map(H.cuda, (p.global_orient, pose, np.concatenate([p.betas, p.expression], axis=1), p.transl)))
else:
    m = H.native_model()
    p = synthetic.make_poses(B, seed=1)
    args = list(map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl)))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    for _ in range(10): m.lbs(*args)
    torch.cuda.synchronize()
ts = []
for _ in range(9):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): m.lbs(*args)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20)
ts.sort()
print(sys.argv[1], B, f"frames: lbs median {ts[4]:.4f} ms (min {ts[0]:.4f}, max {ts[-1]:.4f})")
# parity of the loaded library against the oracle on the first 96 frames (development check of a variant build)
o = H.oracle_model_x() if XMODEL else H.oracle_model()
n = min(B, 96)
fields = ("global_orient", "body_pose", "jaw_pose", "leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose", "betas", "expression",
          "transl") if XMODEL else ("global_orient", "body_pose", "betas", "transl")
with torch.no_grad():
    ref = o(**{k: torch.tensor(getattr(p, k)[:n]) for k in fields})
j, v = m.lbs(*args)
print(f"   max |vertices - oracle| over {n} frames: {float((v[:n].cpu() - ref.vertices).abs().max()):.2e}, joints {float((j[:n].cpu() - ref.joints).abs().max()):.2e}")
