from .frame import optimize_params_frame
from .sequence import optimize_params_sequence, optimize_shape_sequence

__all__ = ["optimize_params_frame", "optimize_params_sequence", "optimize_shape_sequence"]
