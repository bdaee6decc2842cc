"""Development: fit with vertex-selected joints among the targets (k2b_fit_world queues two launches per iteration) vs the fused fit."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from tests import helpers as H
from keypoints2body_amd import native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = dict(np.load(H.GOLDEN / "world_fit_generic_vertex_joints.npz"))
rep = lambda a: H.cuda(np.tile(a, (B // a.shape[0] + 1,) + (1,) * (a.ndim - 1))[:B])
idx = [int(i) for i in d["target_model_indices"]]
kin = [k for k, i in enumerate(idx) if i < 24]
cfg = native.default_fit_config(); cfg.num_iters = 30
args = [rep(d[k]) for k in ("init_global_orient", "init_body_pose", "init_betas", "init_transl")]
j3d, conf = rep(d["j3d"]), H.cuda(d["conf"])
for name, fn in (("with 5 vertex joints", lambda: native.fit_world(H.native_model(), H.native_prior(), cfg, idx, j3d, conf, *args)),
                 ("kinematic only      ", lambda: native.fit_world(H.native_model(), H.native_prior(), cfg, [idx[k] for k in kin], j3d[:, kin].contiguous(), conf[kin].contiguous(), *args))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print(f"{name}: B={B}  {(time.perf_counter() - t0) / 5 * 1e3:8.3f} ms per 30-iteration fit")
