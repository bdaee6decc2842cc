"""Development: LBS forward timed on a steady clock with a given library variant: dev_lbs_loop.py <libname|-> <frames>"""
import sys, time, torch
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
o = H.oracle_model()
n = min(B, 96)
with torch.no_grad():
    ref = o(global_orient=torch.tensor(p.global_orient[:n]), body_pose=torch.tensor(p.body_pose[:n]),
            betas=torch.tensor(p.betas[:n]), transl=torch.tensor(p.transl[:n]))
j, v = m.lbs(*args)
print(f"   max |vertices - oracle| over {n} frames: {float((v[:n].cpu() - ref.vertices).abs().max()):.2e}, joints {float((j[:n].cpu() - ref.joints).abs().max()):.2e}")
