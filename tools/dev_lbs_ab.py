"""Dev: A/B timing of the LBS kernels in ONE process, interleaved rounds (k2b_debug_lbs_kernel: 0 = tile kernel,
1 = the 128 x 64 kernel of round 1), plus agreement between the two and with the float64 twin on a few frames.
    python tools/dev_lbs_ab.py [frames ...]"""
import sys, statistics, torch, numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from keypoints2body_amd import native, synthetic
from tests import helpers as H
lib = native.load_library()
m = H.native_model()
sizes = [int(x) for x in sys.argv[1:]] or [1024, 4096]
for B in sizes:
    p = synthetic.make_poses(B, seed=1)
    args = list(map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl)))
    res, times = {}, {0: [], 1: []}
    for which in (0, 1):
        lib.k2b_debug_lbs_kernel(which)
        for _ in range(3):
            res[which] = m.lbs(*args)
    torch.cuda.synchronize()
    for rnd in range(7):
        for which in (0, 1):
            lib.k2b_debug_lbs_kernel(which)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                m.lbs(*args)
            e1.record(); torch.cuda.synchronize()
            times[which].append(e0.elapsed_time(e1) / 10)
    lib.k2b_debug_lbs_kernel(0)
    dv = (res[0][1] - res[1][1]).abs().max().item(); dj = (res[0][0] - res[1][0]).abs().max().item()
    from oracle.smpl_torch import smpl_forward_np
    e64 = 0.0
    for f in (0, B // 2, B - 1):
        j64, v64 = smpl_forward_np(H.body_consts(), p.global_orient[f], p.body_pose[f], p.betas[f], p.transl[f])
        e64 = max(e64, np.abs(res[0][1][f].cpu().numpy() - v64).max(), np.abs(res[0][0][f].cpu().numpy() - j64).max())
    t0, t1 = statistics.median(times[0]), statistics.median(times[1])
    print(f"B={B}: tile kernel {t0:.4f} ms (min {min(times[0]):.4f}) | 128x64 kernel {t1:.4f} ms (min {min(times[1]):.4f}) | "
          f"speed-up {t1 / t0:.2f}x | tile vs old: verts {dv:.2e} joints {dj:.2e} | tile vs float64 twin {e64:.2e} | "
          f"HBM frac {83560 * B / (t0 * 1e-3) / 8e12:.3f}", flush=True)
