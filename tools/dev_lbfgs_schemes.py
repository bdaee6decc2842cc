"""Development: the three launch schemes of the device L-BFGS on the same frames (persistent / fused rounds / two launches per
round): which frames differ, by how much, and whether each scheme repeats itself."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tests import helpers as H
from tests.test_gpu_lbfgs import _problem
from keypoints2body_amd import native
cus = torch.cuda.get_device_properties(0).multi_processor_count
MI = int(sys.argv[1]) if len(sys.argv) > 1 else 12
j3d, init = _problem(5 * cus, seed=13)
cfg = native.default_fit_config()
run = lambda n: native.fit_world_lbfgs(H.native_model(), H.native_prior(), cfg, list(range(22)), j3d[:n].contiguous(), None,
                                       *[t[:n].contiguous() for t in init], max_iter=MI, lr=1e-2)
sizes = {"persistent": 2 * cus, "fused": 3 * cus, "two-launch": 5 * cus}
names = list(sizes)
if len(sys.argv) > 2:
    torch.save(run(2 * cus)["body_pose"].cpu(), sys.argv[2]); sys.exit(0)
res = {k: run(n) for k, n in sizes.items()}
res2 = {k: run(n) for k, n in sizes.items()}
for k in sizes:
    same = all(torch.equal(res[k][p], res2[k][p]) for p in ("global_orient", "body_pose", "betas", "transl", "loss"))
    print(f"{k:11s} ({sizes[k]} frames): repeats itself {same}")
for i in range(3):
    for j in range(i + 1, 3):
        a, b = res[names[i]], res[names[j]]
        n = min(sizes[names[i]], sizes[names[j]])
        d = (a["body_pose"][:n] - b["body_pose"][:n]).abs().amax(dim=1)
        bad = torch.nonzero(d > 0).flatten()
        print(f"{names[i]} vs {names[j]} on {n} frames: {bad.numel()} frames differ, worst {d.max().item():.2e}, first {bad[:8].tolist()}")

# the same 2 * cus frames under every scheme (K2B_LBFGS_SCHEME is read once per process: one child process per scheme)
if len(sys.argv) <= 2:
    import subprocess, pickle, tempfile
    outs = {}
    for name, env in (("persistent", "0"), ("fused", "2"), ("two-launch", "1")):
        f = tempfile.mktemp(suffix=".pt")
        subprocess.run([sys.executable, __file__, str(MI), f], env=dict(os.environ, K2B_LBFGS_SCHEME=env), check=True, stdout=subprocess.DEVNULL)
        outs[name] = torch.load(f)
    for i in range(3):
        for j in range(i + 1, 3):
            a, b = outs[names[i]], outs[names[j]]
            d = (a - b).abs().amax(dim=1)
            bad = torch.nonzero(d > 0).flatten()
            print(f"SAME {2 * cus} frames, {names[i]} vs {names[j]}: {bad.numel()} frames differ, worst {d.max().item():.2e}, first {bad[:8].tolist()}")
