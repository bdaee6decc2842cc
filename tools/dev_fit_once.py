"""Launch the fused fit a few times at a given batch (for rocprofv3 --pmc passes and quick timing).
usage: python3 tools/dev_fit_once.py FRAMES [pose_prior_weight|-] [launches] [lib variant: tools/libk2b_<name>.so | -] [debug_launch_shape]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from tests import helpers as H
from keypoints2body_amd import native, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wpp = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != '-' else None
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if len(sys.argv) > 4 and sys.argv[4] != '-':
    from pathlib import Path
    native._LIB_PATH = Path(__file__).resolve().parent / f'libk2b_{sys.argv[4]}.so'
m, pr = H.native_model(), H.native_prior()
p = synthetic.make_poses(B, seed=1)
go, bp, be, tr = map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl))
j, _ = m.lbs(go, bp, be, tr, want_vertices=False)
j3d = j[:, :22].contiguous()
z = lambda *s: torch.zeros(*s, device='cuda')
j0, _ = m.lbs(z(B, 3), z(B, 69), z(B, 10), None, want_vertices=False)
tr0 = (j3d[:, 0] - j0[:, 0]).contiguous()
cfg = native.default_fit_config(); cfg.num_iters = int(os.environ.get('K2B_DEV_ITERS', '100'))
if len(sys.argv) > 5: cfg.debug_launch_shape = int(sys.argv[5])
if wpp is not None: cfg.pose_prior_weight = wpp
run = lambda: native.fit_world(m, pr, cfg, list(range(22)), j3d, None, z(B, 3), z(B, 69), z(B, 10), tr0)
o = run(); torch.cuda.synchronize()
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < (0.0 if 'stamp' in ''.join(sys.argv[4:5]) else 0.3):           # device clock ramp (bench.py does the same)
    for _ in range(10): run()
    torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(n): o = run()
ev[1].record(); torch.cuda.synchronize()
print(f'{sys.argv[4] if len(sys.argv) > 4 else "head"} B={B} wpp={wpp}: fit {ev[0].elapsed_time(ev[1]) / n:.4f} ms  loss mean {o["loss"].mean().item():.2f}')
