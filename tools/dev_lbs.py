"""Dev harness: run only the LBS forward a few times (for rocprofv3 PMC passes)."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tests import helpers as H
from keypoints2body_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m = H.native_model()
p = synthetic.make_poses(B, seed=1)
args = list(map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl)))
for _ in range(n):
    j, v = m.lbs(*args)
torch.cuda.synchronize()
print("ok", float(v.abs().mean()))
