"""Dev: per-slice s_memtime stamps of the LBS tile kernel (library built with -DK2B_TILE_DIAG=6):
    python tools/dev_lbs_stamps.py <libname> <frames>
Stamp k of a slice: 0 top, 1 fills issued, 2 compute issued, 3 stores issued, 4 counted wait passed, 5 barrier passed."""
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
for blk, off in ((0, 1024), (77, 3072)):
    st = buf[off:off + 2048].reshape(8, 32, 8).astype(np.int64)
    print(f"block {blk}: cycles (s_memtime) per slice for waves 0 and 7: [issue fills | compute issue | stores | wait vm | barrier] total")
    for w in (0, 7):
        base = st[w, 0, 0]
        for sl in range(11):
            s = st[w, sl]
            if s[0] == 0: continue
            s3 = s[3] if s[3] else s[2]
            nxt = st[w, sl + 1, 0] if st[w, sl + 1, 0] else s[5]
            print(f"  wave {w} slice {sl:2d} @{s[0] - base:7d}: fills {s[1] - s[0]:5d} | compute {s[2] - s[1]:5d} | stores {s3 - s[2]:5d} | wait {s[4] - s3:5d} | barrier {s[5] - s[4]:5d} | to next top {nxt - s[5]:5d} | slice {s[5] - s[0]:6d}")
