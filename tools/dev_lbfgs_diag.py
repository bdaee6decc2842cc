import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests import helpers as H
from tests.test_gpu_lbfgs import _problem, _host_twin
from keypoints2body_amd import native
B = 6
j3d, init = _problem(B, seed=13)
cfg = native.default_fit_config()
cat = lambda o: torch.cat([o[k] for k in ("global_orient", "body_pose", "betas", "transl")], dim=1).cpu().numpy()
for h in (100, 3, 2):
    for it in (3, 4, 5, 6, 8, 10, 12, 16):
        dev = cat(native.fit_world_lbfgs(H.native_model(), H.native_prior(), cfg, list(range(22)), j3d, None, *init, max_iter=it, lr=1e-2, history_size=h))
        twin, rounds = _host_twin(cfg, j3d, init, it, history_size=h)
        d = np.abs(dev - twin).max(axis=1)
        print(f"history {h:3d} max_iter {it:2d}: rounds {rounds:2d}  per-frame max dev", " ".join(f"{x:.1e}" for x in d))
