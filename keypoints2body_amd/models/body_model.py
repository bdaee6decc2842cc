"""Device-resident body model: the object behind the reference's ``model=`` hook.

The reference obtains its body model from ``smplx.create(...)``
(reference ``keypoints2body/api/model_factory.py:19-40``) and only ever calls it with
smplx-style keyword arguments, reading ``.joints`` / ``.vertices`` from the result
(``core/fitters/world_space.py:174-192,259-278``) plus a few optional attributes
(``core/engine.py:140,146,155,180,194-195``).  ``BodyModel`` satisfies that duck
type, but its forward is the HIP LBS kernel (``csrc/k2b_lbs.hip``) and its constant
tensors live in HBM behind a ``k2b_model`` handle that the fused fit kernel shares.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from .. import native, synthetic


# smplx SMPL-X full pose behind the root: body 63 | jaw | left eye | right eye | left hand 45 | right hand 45 (joints 1..54)
SMPLX_POSE_FIELDS = (("body_pose", 63), ("jaw_pose", 3), ("leye_pose", 3), ("reye_pose", 3), ("left_hand_pose", 45),
                     ("right_hand_pose", 45))
SMPLH_POSE_FIELDS = (("body_pose", 63), ("left_hand_pose", 45), ("right_hand_pose", 45))


class BodyModel:
    """SMPL-family model on one MI355X.  No CPU path: construction needs a HIP device.

    55-joint models are SMPL-X (``model_type == "smplx"``), 52-joint models SMPL-H (``"smplh"``): the kernels see ONE pose
    vector of all non-root joints (162 / 153 values) and ONE vector of shape coefficients (betas | expression);
    ``pack_pose`` / ``pack_shape`` / ``unpack`` translate from and to smplx's keyword arguments (hands as full axis-angle
    poses, ``use_pca=False``).  ``packed`` says whether a model takes that route."""

    NUM_BODY_JOINTS = 23
    NUM_HAND_JOINTS = 15

    def __init__(self, v_template, shapedirs, posedirs, J_regressor, lbs_weights, parents,
                 extra_vertex_ids=None, device=None, model_type: str = "smpl", num_betas: Optional[int] = None):
        self.native = native.NativeModel(v_template, shapedirs, posedirs, J_regressor, lbs_weights, parents,
                                         extra_vertex_ids, device=device)
        self.device = self.native.device
        nj = self.native.num_joints
        self.model_type = "smplx" if nj == 55 else ("smplh" if nj == 52 else model_type)
        self.packed = self.model_type in ("smplx", "smplh") and nj in (52, 55)
        self.pose_fields = SMPLX_POSE_FIELDS if nj == 55 else SMPLH_POSE_FIELDS
        self.num_shape = self.native.num_betas                     # everything the shape blend takes
        self.num_expression_coeffs = 0
        if self.packed:
            self.NUM_BODY_JOINTS = 21
            # `shapedirs` = betas | expression: the split is the caller's (smplx: num_betas / num_expression_coeffs of the
            # module); without one, smplx's defaults (10 betas, the rest expression)
            if num_betas is None:
                num_betas = 10 if self.num_shape > 10 else self.num_shape
            if not 1 <= int(num_betas) <= self.num_shape:
                raise ValueError(f"num_betas={num_betas} outside [1, {self.num_shape}]")
            self.num_betas = int(num_betas)
            self.num_expression_coeffs = self.num_shape - self.num_betas
        else:
            self.num_betas = self.num_shape
        self.num_joints = self.native.num_joints
        self.num_vertices = self.native.num_vertices
        self.parents = torch.as_tensor(np.asarray(parents), dtype=torch.long)

    # -- constructors ---------------------------------------------------------------------
    @classmethod
    def synthetic(cls, seed: int = 0, device=None) -> "BodyModel":
        """SMPL-shaped synthetic model (see ``keypoints2body_amd.synthetic``)."""
        c = synthetic.make_body_model(seed)
        return cls(c.v_template, c.shapedirs, c.posedirs, c.J_regressor, c.lbs_weights, c.parents,
                   c.extra_vertex_ids, device=device)

    @classmethod
    def synthetic_h(cls, seed: int = 0, device=None) -> "BodyModel":
        """SMPL-H-shaped synthetic model (52 joints, V = 6890, 10 betas)."""
        c = synthetic.make_body_model_h(seed)
        return cls(c.v_template, c.shapedirs, c.posedirs, c.J_regressor, c.lbs_weights, c.parents,
                   c.extra_vertex_ids, device=device, model_type="smplh")

    @classmethod
    def synthetic_x(cls, seed: int = 0, device=None) -> "BodyModel":
        """SMPL-X-shaped synthetic model (55 joints, V = 10475, 10 betas + 10 expression coefficients)."""
        c = synthetic.make_body_model_x(seed)
        return cls(c.v_template, c.shapedirs, c.posedirs, c.J_regressor, c.lbs_weights, c.parents,
                   c.extra_vertex_ids, device=device, model_type="smplx")

    # -- SMPL-X packing ---------------------------------------------------------------------
    def pack_pose(self, B: int, **kw) -> torch.Tensor:
        """(B, 3 (J - 1)) pose of all non-root joints from smplx keyword arguments (missing ones are zero)."""
        if not self.packed:
            return self._as_dev(kw.get("body_pose"), 3 * (self.num_joints - 1)) if kw.get("body_pose") is not None \
                else torch.zeros((B, 3 * (self.num_joints - 1)), dtype=torch.float32, device=self.device)
        parts = []
        for name, cols in self.pose_fields:
            x = kw.get(name)
            t = self._as_dev(x, cols) if x is not None else torch.zeros((B, cols), dtype=torch.float32, device=self.device)
            parts.append(t.expand(B, -1) if t.shape[0] != B else t)
        return torch.cat(parts, dim=1).contiguous()

    def pack_shape(self, B: int, betas=None, expression=None) -> torch.Tensor:
        z = lambda c: torch.zeros((B, c), dtype=torch.float32, device=self.device)
        if not self.packed:
            return self._as_dev(betas, self.num_shape) if betas is not None else z(self.num_shape)
        be = self._as_dev(betas, self.num_betas) if betas is not None else z(self.num_betas)
        ex = self._as_dev(expression, self.num_expression_coeffs) if expression is not None else z(self.num_expression_coeffs)
        be, ex = (t.expand(B, -1) if t.shape[0] != B else t for t in (be, ex))
        return torch.cat((be, ex), dim=1).contiguous()

    def unpack(self, pose: torch.Tensor, shape: torch.Tensor) -> dict:
        """Inverse of pack_pose / pack_shape: smplx field name -> tensor."""
        if not self.packed:
            return {"body_pose": pose, "betas": shape}
        out, o = {}, 0
        for name, cols in self.pose_fields:
            out[name] = pose[:, o:o + cols].contiguous()
            o += cols
        out["betas"] = shape[:, :self.num_betas].contiguous()
        out["expression"] = shape[:, self.num_betas:].contiguous()
        return out

    @classmethod
    def from_smplx(cls, model, device=None) -> "BodyModel":
        """Take the constants of a loaded ``smplx`` module (or any object exposing
        ``v_template, shapedirs, posedirs, J_regressor, lbs_weights, parents`` and, optionally, ``expr_dirs``,
        ``num_betas``, ``num_expression_coeffs``, ``pose_mean``, ``vertex_joint_selector.extra_joints_idxs``): see
        ``smplx_constants`` for the conventions."""
        return cls(**smplx_constants(model), device=device)

    @classmethod
    def from_npz(cls, path: str, device=None) -> "BodyModel":
        with np.load(path) as z:
            extra = z["extra_vertex_ids"] if "extra_vertex_ids" in z else None
            return cls(z["v_template"], z["shapedirs"], z["posedirs"], z["J_regressor"], z["lbs_weights"],
                       z["parents"], extra, device=device)

    # -- smplx-style forward --------------------------------------------------------------
    def _as_dev(self, x, cols) -> torch.Tensor:
        t = torch.as_tensor(x, dtype=torch.float32).detach()
        if t.dim() == 1:
            t = t.unsqueeze(0)
        if t.shape[-1] != cols:
            raise ValueError(f"expected a (B,{cols}) tensor, got {tuple(t.shape)}")
        return t.to(self.device).contiguous()

    def __call__(self, global_orient=None, body_pose=None, betas=None, transl=None,
                 return_full_pose: bool = False, return_verts: bool = True, **unused):
        extra = {k: unused.get(k) for k in ("jaw_pose", "leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose", "expression")}
        given = [x for x in (global_orient, body_pose, betas, transl, *extra.values()) if x is not None]
        B = max((int(torch.as_tensor(x).reshape(-1, torch.as_tensor(x).shape[-1]).shape[0]) for x in given), default=1)
        zeros = lambda c: torch.zeros((B, c), dtype=torch.float32, device=self.device)
        go = self._as_dev(global_orient, 3) if global_orient is not None else zeros(3)
        if self.packed:
            bp = self.pack_pose(B, body_pose=body_pose, **{k: v for k, v in extra.items() if k != "expression"})
            be = self.pack_shape(B, betas, extra["expression"])
        else:
            D = 3 * (self.num_joints - 1)
            bp = self._as_dev(body_pose, D) if body_pose is not None else zeros(D)
            be = self._as_dev(betas, self.num_betas) if betas is not None else zeros(self.num_betas)
        tr = self._as_dev(transl, 3) if transl is not None else None
        go, bp, be = (t.expand(B, -1).contiguous() if t.shape[0] != B else t for t in (go, bp, be))
        if tr is not None and tr.shape[0] != B:
            tr = tr.expand(B, -1).contiguous()
        joints, verts = self.native.lbs(go, bp, be, tr, want_vertices=return_verts)
        return SimpleNamespace(vertices=verts, joints=joints, betas=be[:, :self.num_betas], global_orient=go,
                               body_pose=bp[:, :3 * self.NUM_BODY_JOINTS],
                               full_pose=torch.cat((go, bp), dim=1) if return_full_pose else None)

    forward = __call__


def smplx_constants(model) -> dict:
    """Constructor arguments of ``BodyModel`` from an smplx-style module (pure attribute reading: no device needed).

    Conventions, as smplx (>= 0.1.28, the reference's pin) lays its buffers out - PARITY UNPINNED at that boundary (smplx is
    not installed here; covered by a stub object in ``tests/test_host_logic.py``):

    * ``shapedirs`` holds the betas' directions (``[:, :, :num_betas]``); SMPL-X keeps the expression directions in a SEPARATE
      buffer ``expr_dirs`` (V, 3, num_expression_coeffs): they are concatenated behind the betas, so the kernels see one
      shape vector betas | expression and ``num_betas`` records the split (16-beta SMPL-H keeps all 16);
    * hands are full axis-angle poses (``use_pca=False``); a module built with ``use_pca=True`` has 6-12 PCA coefficients per
      hand, which this engine does not take (the packed pose has 45 values per hand);
    * ``pose_mean`` (smplx adds it to the full pose before Rodrigues; non-zero for ``flat_hand_mean=False``, the smplx default)
      would make a zero hand pose mean a different mesh than here: a non-zero one is refused - build the smplx module with
      ``flat_hand_mean=True``.
    """
    get = lambda name: getattr(model, name)
    selector = getattr(model, "vertex_joint_selector", None)
    extra = getattr(selector, "extra_joints_idxs", None) if selector is not None else None
    if extra is None:
        extra = getattr(model, "extra_vertex_ids", None)
    host = lambda x: np.asarray(x.detach().cpu() if isinstance(x, torch.Tensor) else x)
    shapedirs = host(get("shapedirs"))
    nb = int(getattr(model, "num_betas", shapedirs.shape[2]))
    nb = min(nb, shapedirs.shape[2])
    shapedirs = shapedirs[:, :, :nb]
    expr = getattr(model, "expr_dirs", None)
    if expr is not None:
        expr = host(expr)
        ne = int(getattr(model, "num_expression_coeffs", expr.shape[2]))
        shapedirs = np.concatenate([shapedirs, expr[:, :, :min(ne, expr.shape[2])]], axis=2)
    if bool(getattr(model, "use_pca", False)):
        raise NotImplementedError("an smplx module with use_pca=True (PCA hand coefficients): build it with use_pca=False, "
                                  "this engine fits full axis-angle hand poses")
    pose_mean = getattr(model, "pose_mean", None)
    if pose_mean is not None and float(np.abs(host(pose_mean)).max()) > 0.0:
        raise NotImplementedError("an smplx module with a non-zero pose_mean (flat_hand_mean=False): build it with "
                                  "flat_hand_mean=True - the kernels apply the pose as given")
    nj = int(host(get("parents")).shape[0])
    return dict(v_template=host(get("v_template")), shapedirs=np.ascontiguousarray(shapedirs), posedirs=host(get("posedirs")),
                J_regressor=host(get("J_regressor")), lbs_weights=host(get("lbs_weights")), parents=host(get("parents")),
                extra_vertex_ids=None if extra is None else host(extra),
                model_type="smplx" if nj == 55 else ("smplh" if nj == 52 else "smpl"), num_betas=nb)


def as_body_model(model, device=None) -> BodyModel:
    """Accept this engine's ``BodyModel`` or adopt an smplx-style module's constants."""
    if isinstance(model, BodyModel):
        return model
    needed = ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "parents")
    if all(hasattr(model, n) for n in needed):
        return BodyModel.from_smplx(model, device=device)
    raise ValueError(
        "model must be a keypoints2body_amd BodyModel or an smplx-style module exposing "
        + ", ".join(needed) + "; an opaque callable cannot be run by the HIP kernels")
