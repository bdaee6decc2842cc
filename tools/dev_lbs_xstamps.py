"""Per-phase s_memtime stamps of the SMPL-X stream kernel (library built with -DK2B_STREAMX_DIAG=2): dev_lbs_xstamps.py <lib> <frames>"""
import sys, ctypes, numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from keypoints2body_amd import native
native._LIB_PATH = Path(__file__).resolve().parent / f"libk2b_{sys.argv[1]}.so"
from tests import helpers as H
from keypoints2body_amd import synthetic
B = int(sys.argv[2])
m = H.native_model_x()
p = synthetic.make_poses_x(B, seed=1)
pose = np.concatenate([getattr(p, k) for k in ("body_pose", "jaw_pose", "leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose")], axis=1)
args = list(map(H.cuda, (p.global_orient, pose, np.concatenate([p.betas, p.expression], axis=1), p.transl)))
for _ in range(30):
    m.lbs(*args)
torch.cuda.synchronize()
buf = np.zeros(16384, np.uint32)
native._check(native.load_library().k2b_debug_read_dump(m.handle, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes), "read_dump")
for blk, off in ((0, 1024), (77, 3072)):
    st = buf[off:off + 512].reshape(8, 64).astype(np.int64)
    t0 = st[:, 0].min()
    print(f"block {blk}: second tile, cycles since the first wave's tile start")
    print("  wave | start | pose k-steps 0..15 (duration) | pose total | per unit: wait+barrier / compute / stores (until the next stamp) | transform total | tile")
    for w in range(8):
        s = st[w]
        ks = [s[i + 1] - s[i] for i in range(16)]
        units = []
        for u in range(8):
            a0, a1, a2 = s[16 + 3 * u], s[17 + 3 * u], s[18 + 3 * u]
            nxt = s[16 + 3 * (u + 1)] if u < 7 else s[40]
            units.append(f"{a1 - a0}/{a2 - a1}/{nxt - a2}")
        print(f"  {w} | {s[0] - t0:6d} | {' '.join(f'{k:5d}' for k in ks)} | {s[16] - s[0]:6d} | {' '.join(units)} | {s[40] - s[16]:6d} | {s[40] - s[0]:6d}")
