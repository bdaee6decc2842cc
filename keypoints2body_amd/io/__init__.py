"""On-disk formats either side of the fitting path (reference ``keypoints2body/io/__init__.py``)."""
from .motion import load_motion_data, write_smplx_zip, write_smplx_zip_from_smpl_data

__all__ = ["load_motion_data", "write_smplx_zip", "write_smplx_zip_from_smpl_data"]
