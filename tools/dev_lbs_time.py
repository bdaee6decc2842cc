"""Dev: time the LBS forward with a given library variant: dev_lbs_time.py <libname|-> <frames>"""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from keypoints2body_amd import native
if sys.argv[1] != "-":
    native._LIB_PATH = Path(__file__).resolve().parent / f"libk2b_{sys.argv[1]}.so"
from tests import helpers as H
from keypoints2body_amd import synthetic
B = int(sys.argv[2])
m = H.native_model()
p = synthetic.make_poses(B, seed=1)
args = list(map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl)))
for _ in range(3): m.lbs(*args)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): m.lbs(*args)
e1.record(); torch.cuda.synchronize()
print(sys.argv[1], B, "frames: lbs", round(e0.elapsed_time(e1) / 10, 4), "ms")
