"""bench.py's rank-spawning contract, checked without a GPU: `python bench.py --gpus N` (N > 1, no launcher environment) must start
its own ranks before touching the device, refuse when fewer devices than ranks are visible, and never fall through to a 1-GPU run
that reports n_gpus = 1 (the driver's N > 1 form would otherwise measure one GPU)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.skipif(torch.cuda.device_count() > 0, reason="checks the behaviour on a host WITHOUT a HIP device")


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_above_visible_devices_is_refused_not_downgraded():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-weak-line"])
    assert r.returncode != 0
    assert '"n_gpus"' not in r.stdout                      # no JSON line of a silently smaller run
    assert "HIP device" in r.stderr or "visible" in r.stderr


def test_single_gpu_run_without_device_fails_loudly():
    r = _run(["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-weak-line"])
    assert r.returncode != 0
    assert "no CPU path" in r.stderr or "HIP device" in r.stderr
    assert '"value"' not in r.stdout
