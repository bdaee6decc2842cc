from .base import BodyEstimator
from .factory import create_estimator
from .optimization import OptimizationEstimator

__all__ = ["BodyEstimator", "OptimizationEstimator", "create_estimator"]
