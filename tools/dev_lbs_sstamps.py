"""Per-phase s_memtime stamps of the LBS stream kernel (library built with -DK2B_STREAM_DIAG=2): dev_lbs_sstamps.py <lib> <frames>"""
import sys, ctypes, numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from keypoints2body_amd import native
native._LIB_PATH = Path(__file__).resolve().parent / f"libk2b_{sys.argv[1]}.so"
from tests import helpers as H
from keypoints2body_amd import synthetic
B = int(sys.argv[2])
m = H.native_model()
p = synthetic.make_poses(B, seed=1)
args = list(map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl)))
for _ in range(30):
    m.lbs(*args)
buf = np.zeros(16384, np.uint32)
native._check(native.load_library().k2b_debug_read_dump(m.handle, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes), "read_dump")
for blk, off in ((0, 1024), (77, 3072)):
    st = buf[off:off + 256].reshape(8, 32).astype(np.int64)
    t0 = st[:, 0].min()
    print(f"block {blk}: third tile, cycles since the first wave's tile start")
    print("  wave | start | pose k-steps 0..6 (duration) | per unit: wait+barrier / compute / stores(until next unit's stamp)")
    for w in range(8):
        s = st[w]
        ks = [s[i + 1] - s[i] for i in range(7)]
        units = []
        for u in range(8):
            a0, a1, a2 = s[8 + 3 * u], s[9 + 3 * u], s[10 + 3 * u]
            nxt = s[8 + 3 * (u + 1)] if u < 7 else a2
            units.append(f"{a1 - a0}/{a2 - a1}/{nxt - a2}")
        print(f"  {w} | {s[0] - t0:6d} | {' '.join(f'{k:5d}' for k in ks)} | pose total {s[7] - s[0]:6d} | {' '.join(units)} | transform total {s[31] - s[8]:6d}")
