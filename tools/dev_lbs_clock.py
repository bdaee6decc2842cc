"""In-kernel clock of the LBS stream kernel under sustained load (library built with -DK2B_STREAM_DIAG=1):
    python tools/dev_lbs_clock.py <libname> <frames>
Prints the median over workgroups of (shader cycles / 100 MHz ticks) x 100 MHz and the kernel's cycles per workgroup."""
import sys, time, ctypes, numpy as np, torch
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
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.0:          # two seconds of back-to-back launches: the clock the chip HOLDS under this load
    for _ in range(20): m.lbs(*args)
    torch.cuda.synchronize()
buf = np.zeros(16384, np.uint32)
native._check(native.load_library().k2b_debug_read_dump(m.handle, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes), "read_dump")
d = buf[1024:1024 + 4 * 256].reshape(256, 4).astype(np.float64)
ok = d[:, 1] > 0
clk = d[ok, 0] / d[ok, 1] * 100e6
print(f"{B} frames: in-kernel clock median {np.median(clk) / 1e9:.3f} GHz (min {clk.min() / 1e9:.3f}, max {clk.max() / 1e9:.3f}) over {ok.sum()} workgroups; "
      f"cycles per workgroup median {np.median(d[ok, 0]):.0f}")
