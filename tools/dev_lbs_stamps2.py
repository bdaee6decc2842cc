"""Dev: all eight waves' stamps of chosen slices (library built with -DK2B_TILE_DIAG=6): dev_lbs_stamps2.py <lib> <frames> <slice> [<slice> ...]"""
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
for _ in range(3):
    m.lbs(*args)
buf = np.zeros(16384, np.uint32)
native._check(native.load_library().k2b_debug_read_dump(m.handle, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes), "read_dump")
st = buf[1024:1024 + 2048].reshape(8, 32, 8).astype(np.int64)
t0 = st[:, 0, 0].min()
for sl in [int(x) for x in sys.argv[3:]]:
    print(f"slice {sl}: per wave, stamps relative to the tile's first stamp: top | +1 | +2 (MFMAs issued) | +3 | +4 (wait passed) | +5 (barrier passed) | +6 entries third done | +7 two thirds")
    for w in (0, 7):
        s = st[w, sl]
        print(f"  wave {w}: " + " ".join(f"{(x - t0) if x else 0:7d}" for x in s[:8]))
