"""keypoints2body_amd — MI355X-native SMPLify-style body fitting.

Drop-in for the hot path of ``keypoints2body`` (per-frame Adam fitting of SMPL parameters to
3D joints): same public functions and data containers, executed by hand-written HIP kernels
through the C ABI in ``include/k2b.h``.  There is no CPU fallback.
"""
from .api.frame import optimize_params_frame
from .api.sequence import optimize_params_sequence, optimize_shape_sequence
from .models.smpl_data import (BodyModelFitResult, BodyModelParams, FLAMEData, MANOData, SMPLData, SMPLHData,
                               SMPLXData)

__version__ = "0.1.0"

__all__ = [
    "__version__", "optimize_params_frame", "optimize_params_sequence", "optimize_shape_sequence",
    "BodyModelFitResult", "BodyModelParams", "MANOData", "FLAMEData", "SMPLData", "SMPLHData", "SMPLXData",
]
