"""Max-mixture pose prior: host-side preparation of the buffers the HIP kernel reads.

Mirrors what the reference's ``MaxMixturePrior.__init__`` derives from a
``gmm_XX.pkl`` mixture (reference ``keypoints2body/core/prior.py:98-176``):
``means``, per-component ``precisions = inv(float32 covariance)`` and
``nll_weights = w / ((2 pi)^(69/2) * sqrt(det) / min sqrt(det))``.  The arithmetic of
the prior itself (``merged_log_likelihood``, prior.py:182-195) runs inside the fused
fit kernel (``csrc/k2b_fit.hip``), not here.
"""
from __future__ import annotations

import io
import os
import pickle
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import native


class _ArraysOnlyUnpickler(pickle.Unpickler):
    """Unpickler for the reference's ``gmm_XX.pkl`` (a dict of numpy arrays, protocol 2, written by Python 2):
    only the globals numpy's own array pickling needs are resolvable, so loading the file cannot run code.
    sklearn ``GMM`` objects (the reference's second accepted form, prior.py:140-143) are refused: unpickling one
    means importing and executing arbitrary classes named by the file."""

    _ALLOWED = {
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
        ("numpy", "ndarray"), ("numpy", "dtype"),
        ("_codecs", "encode"),      # how Python 3 writes an array's bytes into a protocol-2 pickle (a pure str -> bytes function)
    }

    def find_class(self, module, name):
        if (module, name) == ("_codecs", "encode"):
            import codecs
            return codecs.encode
        if (module, name) in self._ALLOWED:
            import numpy.core.multiarray as _ma     # numpy 1.x and 2.x both resolve this path
            if name in ("_reconstruct", "scalar"):
                return getattr(_ma, name)
            return getattr(np, name)
        raise pickle.UnpicklingError(
            f"mixture prior file names the global {module}.{name}; only plain dicts of numpy arrays are loaded "
            "(convert other formats to an .npz with means / covars / weights)")


def _load_mixture_pickle(path: str):
    with open(path, "rb") as f:
        return _ArraysOnlyUnpickler(io.BytesIO(f.read()), encoding="latin1").load()


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
        """Load a user-supplied mixture: ``.npz`` (means/covars/weights) or the reference's ``gmm_XX.pkl``
        dict of arrays (prior.py:133-139), read with an unpickler that resolves numpy array globals only.

        Deliberate differences from the reference: a missing file raises ``FileNotFoundError`` and an
        unsupported content ``ValueError`` where the reference prints and calls ``sys.exit(-1)``
        (prior.py:116-118,126-131,144-146) - a library must not end its host process - and pickled
        sklearn ``GMM`` objects (prior.py:140-143) are refused instead of executed."""
        if not os.path.exists(path):
            raise FileNotFoundError(f'The path to the mixture prior "{path}" does not exist')
        if path.endswith(".npz"):
            with np.load(path) as z:
                return cls.from_mixture(z["means"], z["covars"], z["weights"])
        try:
            gmm = _load_mixture_pickle(path)
        except pickle.UnpicklingError as e:
            raise ValueError(f"cannot load the mixture prior {path}: {e}") from e
        if isinstance(gmm, dict) and all(k in gmm for k in ("means", "covars", "weights")):
            return cls.from_mixture(gmm["means"], gmm["covars"], gmm["weights"])
        raise ValueError(f"Unknown content for the prior {path}: {type(gmm)} (expected a dict with means / covars / weights)")


class MaxMixturePrior:
    """Device-resident prior: owns the native handle the fit kernel consumes."""

    def __init__(self, buffers: Optional[MixtureBuffers] = None, *, prior_folder: str = "./data/models/",
                 num_gaussians: int = 8, device=None):
        if buffers is None:
            buffers = MixtureBuffers.from_file(os.path.join(prior_folder, "gmm_{:02d}.pkl".format(num_gaussians)))
        self.buffers = buffers
        self.num_gaussians = int(buffers.means.shape[0])
        self.native = native.NativePrior(buffers.means, buffers.precisions, buffers.nll_weights, device=device)
