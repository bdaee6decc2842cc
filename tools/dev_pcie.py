"""PCIe-inclusive rate of one step (host buffers in, host buffers out) next to the resident rate.
usage: python3 tools/dev_pcie.py [FRAMES]   -- the number DESIGN.md §5 quotes; never bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from tests import helpers as H
from keypoints2body_amd import native, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m, pr = H.native_model(), H.native_prior()
p = synthetic.make_poses(B, seed=1)
go, bp, be, tr = map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl))
j, _ = m.lbs(go, bp, be, tr, want_vertices=False)
j3d_host = j[:, :22].contiguous().cpu().pin_memory()
tr0_host = (j[:, 0]).cpu().pin_memory()
cfg = native.default_fit_config(); cfg.num_iters = 100
z = lambda *s: torch.zeros(*s, device='cuda')
V = m.num_vertices
out_host = {k: torch.empty(s).pin_memory() for k, s in
            dict(go=(B, 3), bp=(B, 69), be=(B, 10), tr=(B, 3), joints=(B, 24 + m.num_extra, 3), verts=(B, V, 3)).items()}

def step(host, verts=True):
    if host:
        j3d = j3d_host.to('cuda', non_blocking=True); tr0 = tr0_host.to('cuda', non_blocking=True)
    else:
        j3d, tr0 = step.j3d, step.tr0
    o = native.fit_world(m, pr, cfg, list(range(22)), j3d, None, z(B, 3), z(B, 69), z(B, 10), tr0)
    jo, vo = m.lbs(o['global_orient'], o['body_pose'], o['betas'], o['transl'], want_vertices=True)
    if host:
        for k, t in (('go', o['global_orient']), ('bp', o['body_pose']), ('be', o['betas']), ('tr', o['transl']), ('joints', jo)):
            out_host[k].copy_(t, non_blocking=True)
        if verts:
            out_host['verts'].copy_(vo, non_blocking=True)
step.j3d, step.tr0 = j3d_host.cuda(), tr0_host.cuda()

def rate(n=50, **kw):
    for _ in range(10): step(**kw)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): step(**kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    return dt * 1e3, B / dt

for name, kw in (('resident', dict(host=False)), ('host buffers, parameters+joints back', dict(host=True, verts=False)),
                 ('host buffers, vertices back too', dict(host=True, verts=True))):
    ms, fps = rate(**kw)
    print(f'{B} frames, {name}: {ms:.3f} ms/step, {fps / 1e6:.3f} M frames/s')
