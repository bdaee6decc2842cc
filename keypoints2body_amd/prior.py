"""Max-mixture pose prior: host-side preparation of the buffers the HIP kernel reads.

Mirrors what the reference's ``MaxMixturePrior.__init__`` derives from a
``gmm_XX.pkl`` mixture (reference ``keypoints2body/core/prior.py:98-176``):
``means``, per-component ``precisions = inv(float32 covariance)`` and
``nll_weights = w / ((2 pi)^(69/2) * sqrt(det) / min sqrt(det))``.  The arithmetic of
the prior itself (``merged_log_likelihood``, prior.py:182-195) runs inside the fused
fit kernel (``csrc/k2b_fit.hip``), not here.
"""
from __future__ import annotations

import os
import pickle
import sys
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import native


@dataclass
class MixtureBuffers:
    """float32 buffers with the names the reference registers on its module."""

    means: np.ndarray         # (M, D)
    precisions: np.ndarray    # (M, D, D)
    nll_weights: np.ndarray   # (M,)

    @classmethod
    def from_mixture(cls, means, covars, weights) -> "MixtureBuffers":
        means = np.asarray(means)
        covars = np.asarray(covars)
        weights = np.asarray(weights)
        if means.ndim != 2 or covars.shape != (means.shape[0], means.shape[1], means.shape[1]) \
                or weights.shape != (means.shape[0],):
            raise ValueError(
                f"mixture arrays disagree: means {means.shape}, covars {covars.shape}, weights {weights.shape}")
        covs32 = covars.astype(np.float32)
        precisions = np.stack([np.linalg.inv(c) for c in covs32]).astype(np.float32)   # prior.py:150-151
        sqrdets = np.array([np.sqrt(np.linalg.det(c)) for c in covars])                # prior.py:156
        const = (2 * np.pi) ** (69 / 2.0)                                              # prior.py:157 (69 is literal there)
        nll = np.asarray(weights / (const * (sqrdets / sqrdets.min())))                # prior.py:159
        return cls(means.astype(np.float32), precisions, nll.astype(np.float32).reshape(-1))

    @classmethod
    def from_file(cls, path: str) -> "MixtureBuffers":
        """Load a user-supplied mixture: ``.npz`` (means/covars/weights) or the
        reference's ``gmm_XX.pkl`` dict / sklearn object (prior.py:133-146)."""
        if not os.path.exists(path):
            # the reference prints and calls sys.exit(-1) here (prior.py:126-131)
            print(f'The path to the mixture prior "{path}" does not exist, exiting!')
            sys.exit(-1)
        if path.endswith(".npz"):
            with np.load(path) as z:
                return cls.from_mixture(z["means"], z["covars"], z["weights"])
        with open(path, "rb") as f:
            gmm = pickle.load(f, encoding="latin1")
        if isinstance(gmm, dict):
            return cls.from_mixture(gmm["means"], gmm["covars"], gmm["weights"])
        if "sklearn.mixture.gmm.GMM" in str(type(gmm)):
            return cls.from_mixture(gmm.means_, gmm.covars_, gmm.weights_)
        print(f"Unknown type for the prior: {type(gmm)}, exiting!")
        sys.exit(-1)


class MaxMixturePrior:
    """Device-resident prior: owns the native handle the fit kernel consumes."""

    def __init__(self, buffers: Optional[MixtureBuffers] = None, *, prior_folder: str = "./data/models/",
                 num_gaussians: int = 8, device=None):
        if buffers is None:
            buffers = MixtureBuffers.from_file(os.path.join(prior_folder, "gmm_{:02d}.pkl".format(num_gaussians)))
        self.buffers = buffers
        self.num_gaussians = int(buffers.means.shape[0])
        self.native = native.NativePrior(buffers.means, buffers.precisions, buffers.nll_weights, device=device)
