"""Golden vectors for the rows either side of the fitting path: motion-file I/O and the MPJAE metric.

Runs only in the build container: it imports the REAL reference modules
``keypoints2body/io/motion.py`` and ``keypoints2body/cli/eval.py`` from /root/reference (package roots
stubbed so that h5py / smplx are never imported) on seeded synthetic inputs, and commits inputs and
outputs as ``tests/golden/io_motion.npz`` and ``tests/golden/mpjae.npz``.
"""
from __future__ import annotations

import sys
import tempfile
import types
import zipfile
import io as _io
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
REF_ROOT = Path("/root/reference/keypoints2body")
GOLDEN = REPO / "tests" / "golden"


def import_reference():
    for name, sub in (("keypoints2body", ""), ("keypoints2body.core", "core"), ("keypoints2body.cli", "cli"),
                      ("keypoints2body.io", "io")):
        mod = types.ModuleType(name)
        mod.__path__ = [str(REF_ROOT / sub)]
        sys.modules[name] = mod
    sys.modules["keypoints2body"].optimize_params_sequence = None      # imported by cli/eval.py, never called here
    from keypoints2body.io import motion  # type: ignore
    from keypoints2body.cli import eval as ev  # type: ignore
    return motion, ev


def capture_csv_text(joints: np.ndarray) -> str:
    """A motion-capture style export: five header rows, frame index and time stamp, then x,y,z per joint."""
    T, K, _ = joints.shape
    head = ["Format Version,1.23,Take Name,synthetic", "", ",,Type" + ",Bone" * (3 * K), ",,Name" + "".join(f",j{k}" * 3 for k in range(K)),
            "Frame,Time (Seconds)" + ",X,Y,Z" * K]
    rows = [f"{t},{t / 120.0:.6f}," + ",".join(repr(float(v)) for v in joints[t].reshape(-1)) for t in range(T)]
    return "\n".join(head + rows) + "\n"


def main():
    motion, ev = import_reference()
    rng = np.random.default_rng(2024)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        # ---- load_motion_data -------------------------------------------------------------------------
        j22 = rng.normal(0, 0.4, (7, 22, 3))
        j24 = rng.normal(0, 0.4, (5, 24, 3)).astype(np.float32)
        np.save(tmp / "a22.npy", j22)
        np.savez(tmp / "b24.npz", joints=j24, other=np.zeros(3))
        csv_text = capture_csv_text(np.round(rng.normal(0, 0.4, (4, 22, 3)), 6))
        (tmp / "c22.csv").write_text(csv_text)
        for tag, fname, layout in (("npy22", "a22.npy", None), ("npz24", "b24.npz", None), ("csv22", "c22.csv", None),
                                   ("npy22_explicit", "a22.npy", "AMASS")):
            joints, lay, k = motion.load_motion_data(tmp / fname, layout)
            out[f"{tag}_joints"], out[f"{tag}_layout"], out[f"{tag}_k"] = np.asarray(joints), str(lay), int(k)
        out["in_j22"], out["in_j24"], out["in_csv_text"] = j22, j24, np.frombuffer(csv_text.encode(), dtype=np.uint8)
        # ---- write_smplx_zip --------------------------------------------------------------------------
        poses = rng.normal(0, 0.3, (3, 72))
        betas = rng.normal(0, 1.0, (10,)).astype(np.float32)
        transl = rng.normal(0, 1.0, (3, 3))
        zp = motion.write_smplx_zip(tmp, poses, betas, transl, zip_name="seq.zip", person_idx=2)
        with zipfile.ZipFile(zp) as zf:
            names = zf.namelist()
            out["zip_names"] = np.array(names)
            for n in names:
                with np.load(_io.BytesIO(zf.read(n))) as d:
                    out["zip_keys"] = np.array(list(d.keys()))
                    for k in d.keys():
                        out[f"zip::{n}::{k}"] = d[k]
        out["zip_poses"], out["zip_betas"], out["zip_transl"] = poses, betas, transl
    np.savez_compressed(GOLDEN / "io_motion.npz", **out)

    # ---- MPJAE metric ---------------------------------------------------------------------------------
    T = 48
    gt = rng.normal(0, 0.5, (T, 72)).astype(np.float32)
    pred = (gt + rng.normal(0, 0.05, (T, 72))).astype(np.float32)
    pred[0] = gt[0]                         # identical rotations: the clip at 1 - 1e-6 decides
    pred[1, :6] = 0.0; gt[1, :6] = 0.0      # zero vectors: Taylor branch
    pred[2, :3] = [1e-9, -2e-9, 3e-9]       # below eps
    gt[3, :3] = [np.pi, 0.0, 0.0]; pred[3, :3] = [-np.pi + 1e-3, 0.0, 0.0]    # near the antipode
    pred[4] = -gt[4]                        # large errors
    ang = ev.compute_angular_error_deg(pred.reshape(T, 24, 3), gt.reshape(T, 24, 3))
    mean, total, count = ev.evaluate_pose_pair(pred, gt)
    # ragged call: more predicted frames and a 66-wide ground truth (body only, 22 rotations)
    mean2, total2, count2 = ev.evaluate_pose_pair(np.concatenate([pred, pred[:5]]), gt[:, :66])
    np.savez_compressed(GOLDEN / "mpjae.npz", pred=pred, gt=gt, angles_deg=ang, mean=mean, total=total, count=count,
                        ragged=np.array([mean2, total2, count2]), rotmat_gt=ev.rotvec_to_rotmat(gt.reshape(T, 24, 3)))
    print("io_motion.npz:", sorted(k for k in out if not k.startswith("zip::"))[:12], "... mpjae mean", mean, count)


if __name__ == "__main__":
    main()
