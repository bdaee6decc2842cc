import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from tests import helpers as H
from keypoints2body_amd import native, synthetic
m, pr = H.native_model(), H.native_prior()
for case in H.WORLD_CASES:
    d = H.load_case(case)
    out = H.native_fit(d)
    e = max(np.abs(out[k].cpu().numpy() - d['out_'+k]).max() for k in ('global_orient','body_pose','betas','transl'))
    print(case, 'final max abs param diff vs reference: %.2e' % e)
for B in (1024, 4096, 16384):
    p = synthetic.make_poses(B, seed=1)
    go, bp, be, tr = map(H.cuda, (p.global_orient, p.body_pose, p.betas, p.transl))
    j, _ = m.lbs(go, bp, be, tr, want_vertices=False)
    j3d = j[:, :22].contiguous()
    z = lambda *s: torch.zeros(*s, device='cuda')
    j0, _ = m.lbs(z(B,3), z(B,69), z(B,10), None, want_vertices=False)
    tr0 = (j3d[:,0] - j0[:,0]).contiguous()
    cfg = native.default_fit_config(); cfg.num_iters = 100
    def run():
        o = native.fit_world(m, pr, cfg, list(range(22)), j3d, None, z(B,3), z(B,69), z(B,10), tr0)
        return o
    o = run(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for _ in range(5): o = run()
    ev[1].record()
    for _ in range(5): jj, vv = m.lbs(o['global_orient'], o['body_pose'], o['betas'], o['transl'])
    ev[2].record(); torch.cuda.synchronize()
    tf, tl = ev[0].elapsed_time(ev[1])/5, ev[1].elapsed_time(ev[2])/5
    err = (jj[:, :22] - j3d).norm(dim=-1).mean().item()
    print(f'B={B}: fit {tf:.3f} ms, lbs {tl:.3f} ms -> {B/(tf+tl)*1e3:.0f} frames/s; mean joint err {err*100:.2f} cm; loss mean {o["loss"].mean().item():.1f}')
