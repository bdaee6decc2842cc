"""Phase stamps of the pose set-up kernel: needs tools/libk2b_pstamp.so (tools/build_lbs_variants.sh pstamp:"-DK2B_POSE_STAMPS=1").
usage: python3 tools/dev_pose_stamps.py [frames]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pathlib import Path
import torch, numpy as np
from tests import helpers as H
from keypoints2body_amd import native, synthetic
native._LIB_PATH = Path(__file__).resolve().parent / 'libk2b_pstamp.so'
m = H.native_model()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
p = synthetic.make_poses(B, seed=1)
go, bp, be, tr = map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl))
for _ in range(30): j, v = m.lbs(go, bp, be, tr, want_vertices=True)
torch.cuda.synchronize()
j, _ = m.lbs(go, bp, be, tr, want_vertices=False)
torch.cuda.synchronize()
st = j[:, 0:3].reshape(B, 9).cpu().numpy()
names = ["start", "loads+J(beta)", "rodrigues+barrier", "offset+barrier+X staging", "compose chain", "A staging+barrier", "X stores", "A stores", "vmcnt(0)"]
print("pose set-up stamps (s_memtime, shader cycles), median / p90 over", B, "frames")
for i in range(1, 9):
    d = st[:, i] - st[:, i - 1]
    print(f"  {names[i]:28s} {np.median(d):8.0f} {np.percentile(d, 90):8.0f}")
print(f"  total                        {np.median(st[:, 8]):8.0f} {np.percentile(st[:, 8], 90):8.0f}")
