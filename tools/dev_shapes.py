"""Every launch shape of the fused fit kernel (k2b_fit_config.debug_launch_shape 1..6): parity against the goldens, bit-identity
against shape 1 on one batch, and time per launch at a list of batch sizes.
usage: python3 tools/dev_shapes.py [frames,frames,...] [shapes e.g. 1,4] [launches]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from tests import helpers as H
from keypoints2body_amd import native, synthetic

NAMES = {0: 'auto', 1: 'split', 2: 'split_paired', 3: 'paired', 4: 'wide'}
CAP = {1: 4, 2: 8, 3: 16, 4: 16}
sizes = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [512, 1024, 2048, 4096]
shapes = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 3, 4]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20

for shape in shapes:
    worst = 0.0
    for case in H.WORLD_CASES:
        d = H.load_case(case)
        cfg_shape = {v: k for k, v in H.LAUNCH_SHAPES.items()}
        from keypoints2body_amd import native as nv
        cfg = nv.default_fit_config()
        cfg.debug_launch_shape = shape
        cfg.num_iters = int(d["num_iters"])
        cfg.pose_preserve_weight = 5.0 if int(d["seq_ind"]) > 0 else 0.0
        cfg.freeze_betas = int(d["freeze_betas"])
        conf = H.cuda(d["conf"]) if int(d["has_conf"]) else None
        out = nv.fit_world(H.native_model(), H.native_prior(), cfg, H.case_indices(d), H.cuda(d["j3d"]), conf,
                           H.cuda(d["init_global_orient"]), H.cuda(d["init_body_pose"]), H.cuda(d["init_betas"]), H.cuda(d["init_transl"]))
        e = max(np.abs(out[k].cpu().numpy() - d['out_' + k]).max() for k in ('global_orient', 'body_pose', 'betas', 'transl'))
        worst = max(worst, e)
    print(f'shape {shape} {NAMES[shape]}: worst |param diff| vs reference goldens: {worst:.2e}', flush=True)

m, pr = H.native_model(), H.native_prior()
def problem(B):
    p = synthetic.make_poses(B, seed=1)
    go, bp, be, tr = map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl))
    j, _ = m.lbs(go, bp, be, tr, want_vertices=False)
    j3d = j[:, :22].contiguous()
    z = lambda *s: torch.zeros(*s, device='cuda')
    j0, _ = m.lbs(z(B, 3), z(B, 69), z(B, 10), None, want_vertices=False)
    return j3d, (j3d[:, 0] - j0[:, 0]).contiguous()

def run(B, shape, j3d, tr0):
    cfg = native.default_fit_config(); cfg.num_iters = 100; cfg.debug_launch_shape = shape
    z = lambda *s: torch.zeros(*s, device='cuda')
    return native.fit_world(m, pr, cfg, list(range(22)), j3d, None, z(B, 3), z(B, 69), z(B, 10), tr0)

# bit identity on a small batch
j3d, tr0 = problem(64)
ref = run(64, 1, j3d, tr0)
for shape in shapes:
    o = run(64, shape, j3d, tr0)
    same = all(torch.equal(o[k], ref[k]) for k in ('global_orient', 'body_pose', 'betas', 'transl', 'loss'))
    dev = max((o[k] - ref[k]).abs().max().item() for k in ('global_orient', 'body_pose', 'betas', 'transl'))
    print(f'shape {shape} {NAMES[shape]} vs split on 64 frames: bit-identical {same}, max dev {dev:.2e}', flush=True)

t0 = time.perf_counter()
j3d, tr0 = problem(4096)
while time.perf_counter() - t0 < 0.4:
    for _ in range(10): run(4096, 0, j3d, tr0)
    torch.cuda.synchronize()
for B in sizes:
    j3d, tr0 = problem(B)
    line = f'B={B:6d}:'
    for shape in [0] + shapes:
        if shape and B > CAP[shape] * 256:
            line += f'  {NAMES[shape]} -'
            continue
        for _ in range(5): run(B, shape, j3d, tr0)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(n): o = run(B, shape, j3d, tr0)
        ev[1].record(); torch.cuda.synchronize()
        line += f'  {NAMES[shape]} {ev[0].elapsed_time(ev[1]) / n:.4f}'
    print(line, flush=True)
