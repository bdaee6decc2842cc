"""Joint-index tables of the fitting path.

The numbering is the SMPL kinematic-tree numbering the reference uses for its
``SMPL24`` and ``AMASS`` joint categories (reference ``keypoints2body/core/constants.py:1-71``):
AMASS observes the first 22 kinematic joints, SMPL24 all 24; SMPL-X block inputs address
model joints 25-45 (left hand), 46-66 (right hand) and 67+ (face).
"""
from __future__ import annotations

_SMPL_JOINT_NAMES = (
    "MidHip", "LHip", "RHip", "spine1", "LKnee", "RKnee", "spine2", "LAnkle", "RAnkle", "spine3",
    "LFoot", "RFoot", "Neck", "LCollar", "Rcollar", "Head", "LShoulder", "RShoulder", "LElbow",
    "RElbow", "LWrist", "RWrist", "LHand", "RHand",
)
_EXTRA_NAMED = {"Nose": 24, "REye": 25, "LEye": 26, "REar": 27, "LEar": 28, "LHeel": 31, "RHeel": 34}

JOINT_MAP = {name: i for i, name in enumerate(_SMPL_JOINT_NAMES)}
JOINT_MAP.update(_EXTRA_NAMED)
AMASS_JOINT_MAP = {name: i for i, name in enumerate(_SMPL_JOINT_NAMES[:22])}

SMPL_IDX = range(24)
AMASS_IDX = range(22)
AMASS_SMPL_IDX = range(22)

SMPLX_BODY_IDX = range(22)
SMPLX_LEFT_HAND_IDX = range(25, 46)
SMPLX_RIGHT_HAND_IDX = range(46, 67)
SMPLX_FACE_IDX_START = 67

# torso joints used by the camera-space initialisation / stage-1 loss
TORSO_JOINTS = ("RHip", "LHip", "RShoulder", "LShoulder")


def category_indices(joints_category: str):
    """(model joint indices, target indices) of a joint category
    (reference ``core/fitters/world_space.py:75-86``)."""
    if joints_category == "SMPL24":
        return list(SMPL_IDX), list(SMPL_IDX)
    if joints_category == "AMASS":
        return list(AMASS_SMPL_IDX), list(AMASS_IDX)
    if joints_category == "GENERIC":
        return None, None
    raise ValueError("No such joints category!")


def root_indices(joints_category: str):
    """(model root index, target root index) used for root alignment
    (reference ``core/fitters/world_space.py:39-47``)."""
    if joints_category == "SMPL24":
        return JOINT_MAP["MidHip"], JOINT_MAP["MidHip"]
    if joints_category == "AMASS":
        return AMASS_JOINT_MAP["MidHip"], AMASS_JOINT_MAP["MidHip"]
    raise ValueError(f"Unknown joints category: {joints_category}")
