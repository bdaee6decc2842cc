"""ctypes binding of ``libk2b.so`` (C ABI: ``include/k2b.h``).

PyTorch is used here only for device memory and streams.  There is no CPU
fallback: if the library is missing, or no HIP device is visible, every entry
point raises ``RuntimeError`` (the product path must fail loudly).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional, Sequence

import numpy as np
import torch

_LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libk2b.so"
_lib = None

K2B_OK = 0
K2B_ERR_INVALID_ARGUMENT = -1
K2B_ERR_UNSUPPORTED = -2
K2B_ERR_HIP = -3
K2B_ERR_NO_DEVICE = -4

EXPORTED_SYMBOLS = (
    "k2b_version", "k2b_last_error", "k2b_model_create", "k2b_model_destroy", "k2b_model_dims",
    "k2b_model_joint_basis", "k2b_model_reserve", "k2b_debug_read_dump", "k2b_prior_create", "k2b_prior_destroy", "k2b_fit_config_default", "k2b_fit_config_size",
    "k2b_fit_world", "k2b_fit_sequence", "k2b_lbs", "k2b_vertex_term", "k2b_adam_step", "k2b_angular_error_deg",
    "k2b_fit_world_lbfgs", "k2b_fit_sequence_lbfgs",
)


class FitConfigC(C.Structure):
    """Mirror of ``k2b_fit_config`` (include/k2b.h)."""

    _fields_ = [
        ("num_iters", C.c_int32),
        ("step_size", C.c_double),
        ("adam_beta1", C.c_double),
        ("adam_beta2", C.c_double),
        ("adam_eps", C.c_double),
        ("sigma", C.c_float),
        ("joint_loss_weight", C.c_float),
        ("pose_prior_weight", C.c_float),
        ("angle_prior_weight", C.c_float),
        ("shape_prior_weight", C.c_float),
        ("pose_preserve_weight", C.c_float),
        ("freeze_betas", C.c_int32),
        ("conf_per_frame", C.c_int32),
        ("angle_prior_index", C.c_int32 * 4),
        ("angle_prior_sign", C.c_float * 4),
        ("optimize_mask", C.c_int32),
        ("transl_prior_weight", C.c_float),
        ("debug_launch_shape", C.c_int32),
        ("prior_pose_dims", C.c_int32),
        ("num_betas_prior", C.c_int32),
    ]


def library_path() -> Path:
    return _LIB_PATH


def load_library():
    """dlopen ``libk2b.so`` and declare prototypes (no device call is made)."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise RuntimeError(
            f"{_LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C keypoints2body_amd/csrc). "
            "keypoints2body_amd has no CPU fallback."
        )
    lib = C.CDLL(str(_LIB_PATH))
    vp, fp, ip = C.c_void_p, C.c_void_p, C.c_void_p
    lib.k2b_version.restype = C.c_uint32
    lib.k2b_last_error.restype = C.c_char_p
    lib.k2b_model_create.restype = C.c_int
    lib.k2b_model_create.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     fp, fp, fp, fp, fp, ip, ip]
    lib.k2b_model_destroy.restype = None
    lib.k2b_model_destroy.argtypes = [vp]
    lib.k2b_model_dims.restype = C.c_int
    lib.k2b_model_dims.argtypes = [vp] + [C.POINTER(C.c_int32)] * 4
    lib.k2b_debug_read_dump.restype = C.c_int
    lib.k2b_debug_read_dump.argtypes = [vp, vp, C.c_int64]
    lib.k2b_model_reserve.restype = C.c_int
    lib.k2b_model_reserve.argtypes = [vp, C.c_int32]
    lib.k2b_model_joint_basis.restype = C.c_int
    lib.k2b_model_joint_basis.argtypes = [vp, fp, fp]
    lib.k2b_prior_create.restype = C.c_int
    lib.k2b_prior_create.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, fp, fp, fp]
    lib.k2b_prior_destroy.restype = None
    lib.k2b_prior_destroy.argtypes = [vp]
    lib.k2b_fit_config_default.restype = None
    lib.k2b_fit_config_default.argtypes = [C.POINTER(FitConfigC)]
    lib.k2b_fit_config_size.restype = C.c_uint32
    if lib.k2b_fit_config_size() != C.sizeof(FitConfigC):
        raise RuntimeError("libk2b.so was built with a different k2b_fit_config layout than native.FitConfigC")
    lib.k2b_fit_world.restype = C.c_int
    lib.k2b_fit_world.argtypes = [vp, vp, C.POINTER(FitConfigC), C.c_int32, C.c_int32, ip] + [fp] * 14 + [vp]
    lib.k2b_fit_sequence.restype = C.c_int
    lib.k2b_fit_sequence.argtypes = [vp, vp, C.POINTER(FitConfigC)] + [C.c_int32] * 4 + [ip] + [fp] * 11 + [vp]
    lib.k2b_lbs.restype = C.c_int
    lib.k2b_lbs.argtypes = [vp, C.c_int32] + [fp] * 6 + [vp]
    lib.k2b_fit_world_lbfgs.restype = C.c_int
    lib.k2b_fit_world_lbfgs.argtypes = ([vp, vp, C.POINTER(FitConfigC), C.c_int32, C.c_int32, ip] + [fp] * 14 +
                                        [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, vp])
    lib.k2b_fit_sequence_lbfgs.restype = C.c_int
    lib.k2b_fit_sequence_lbfgs.argtypes = ([vp, vp, C.POINTER(FitConfigC), C.c_int32, C.c_int32, ip] + [fp] * 11 +
                                           [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, vp])
    lib.k2b_vertex_term.restype = C.c_int
    lib.k2b_vertex_term.argtypes = [vp, C.c_int32, C.c_int32, ip, fp, fp, C.c_float, C.c_float] + [fp] * 6 + [vp]
    lib.k2b_adam_step.restype = C.c_int
    lib.k2b_adam_step.argtypes = [C.c_int64, fp, fp, fp, fp, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, vp]
    lib.k2b_angular_error_deg.restype = C.c_int
    lib.k2b_angular_error_deg.argtypes = [C.c_int64, fp, fp, fp, vp]
    _lib = lib
    return lib


def _check(code: int, what: str):
    if code == K2B_OK:
        return
    msg = load_library().k2b_last_error().decode("utf-8", "replace")
    text = f"{what}: {msg}"
    if code == K2B_ERR_INVALID_ARGUMENT:
        raise ValueError(text)
    if code == K2B_ERR_UNSUPPORTED:
        raise NotImplementedError(text)
    raise RuntimeError(text)


def _host_f32(a) -> np.ndarray:
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def _host_i32(a) -> np.ndarray:
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.int32)


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _dev(t: Optional[torch.Tensor], name: str, device: torch.device, shape=None):
    """Validated device pointer of a contiguous float32 tensor (None -> NULL)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous float32 torch tensor")
    if t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return C.c_void_p(t.data_ptr())


def require_device(device=None) -> torch.device:
    """Resolve a HIP device or raise: the engine has no CPU path."""
    if not torch.cuda.is_available():
        raise RuntimeError("keypoints2body_amd needs a HIP device (MI355X / gfx950); none is visible and there is no CPU fallback")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if dev.type != "cuda":
        raise RuntimeError(f"keypoints2body_amd runs on HIP devices only, got device={dev}")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


class NativeModel:
    """Owner of a ``k2b_model`` handle (body-model constants in HBM)."""

    def __init__(self, v_template, shapedirs, posedirs, J_regressor, lbs_weights, parents,
                 extra_vertex_ids, device=None):
        self.device = require_device(device)
        lib = load_library()
        vt = _host_f32(v_template)
        sd = _host_f32(shapedirs)
        pd = _host_f32(posedirs)
        jr = _host_f32(J_regressor)
        lw = _host_f32(lbs_weights)
        par = _host_i32(parents).copy()
        par[0] = -1      # smplx stores the root's parent as a huge unsigned sentinel
        ex = _host_i32(extra_vertex_ids if extra_vertex_ids is not None else np.zeros(0, np.int32))
        V, J, NB, E = vt.shape[0], par.shape[0], sd.shape[2], ex.shape[0]
        if vt.shape != (V, 3) or sd.shape != (V, 3, NB) or pd.shape != (9 * (J - 1), 3 * V) \
                or jr.shape != (J, V) or lw.shape != (V, J):
            raise ValueError(
                f"inconsistent body-model constants: v_template {vt.shape}, shapedirs {sd.shape}, posedirs {pd.shape}, "
                f"J_regressor {jr.shape}, lbs_weights {lw.shape}, parents {par.shape}")
        self.num_vertices, self.num_joints, self.num_betas, self.num_extra = V, J, NB, E
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib.k2b_model_create(C.byref(self._h), V, J, NB, E, _np_ptr(vt), _np_ptr(sd), _np_ptr(pd),
                                        _np_ptr(jr), _np_ptr(lw), _np_ptr(par), _np_ptr(ex)), "k2b_model_create")

    @property
    def handle(self):
        return self._h

    def reserve(self, max_frames: int) -> None:
        """Pre-size the LBS workspace: no later ``lbs`` call of up to `max_frames` frames allocates or synchronises."""
        with torch.cuda.device(self.device):
            _check(load_library().k2b_model_reserve(self._h, int(max_frames)), "k2b_model_reserve")

    def joint_basis(self):
        jt = np.zeros((self.num_joints, 3), np.float32)
        jd = np.zeros((self.num_joints, 3, self.num_betas), np.float32)
        _check(load_library().k2b_model_joint_basis(self._h, _np_ptr(jt), _np_ptr(jd)), "k2b_model_joint_basis")
        return jt, jd

    def lbs(self, global_orient, body_pose, betas, transl=None, want_vertices=True):
        """Full forward: returns (joints (B,J+E,3), vertices (B,V,3) or None)."""
        dev = self.device
        B = global_orient.shape[0]
        D = 3 * (self.num_joints - 1)
        go = _dev(global_orient, "global_orient", dev, (B, 3))
        bp = _dev(body_pose, "body_pose", dev, (B, D))
        be = _dev(betas, "betas", dev, (B, self.num_betas))
        tr = _dev(transl, "transl", dev, (B, 3))
        joints = torch.empty((B, self.num_joints + self.num_extra, 3), dtype=torch.float32, device=dev)
        verts = torch.empty((B, self.num_vertices, 3), dtype=torch.float32, device=dev) if want_vertices else None
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _check(load_library().k2b_lbs(self._h, B, go, bp, be, tr, C.c_void_p(joints.data_ptr()),
                                          C.c_void_p(verts.data_ptr()) if verts is not None else None, stream), "k2b_lbs")
        return joints, verts

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value and _lib is not None:
                _lib.k2b_model_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class NativePrior:
    """Owner of a ``k2b_prior`` handle built from the reference prior's buffers."""

    def __init__(self, means, precisions, nll_weights, device=None):
        self.device = require_device(device)
        mu = _host_f32(means)
        pr = _host_f32(precisions)
        nw = _host_f32(nll_weights).reshape(-1)
        M, D = mu.shape
        if pr.shape != (M, D, D) or nw.shape != (M,):
            raise ValueError(f"prior buffers disagree: means {mu.shape}, precisions {pr.shape}, nll_weights {nw.shape}")
        self.num_gaussians, self.dim = M, D
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(load_library().k2b_prior_create(C.byref(self._h), M, D, _np_ptr(mu), _np_ptr(pr), _np_ptr(nw)),
                   "k2b_prior_create")

    @property
    def handle(self):
        return self._h

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value and _lib is not None:
                _lib.k2b_prior_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


def default_fit_config() -> FitConfigC:
    cfg = FitConfigC()
    load_library().k2b_fit_config_default(C.byref(cfg))
    return cfg


def fit_world(model: NativeModel, prior: NativePrior, cfg: FitConfigC, model_joint_index: Sequence[int],
              j3d: torch.Tensor, conf: Optional[torch.Tensor], global_orient: torch.Tensor, body_pose: torch.Tensor,
              betas: torch.Tensor, transl: torch.Tensor, preserve_pose: Optional[torch.Tensor] = None,
              want_grad: bool = False, transl_prior_target: Optional[torch.Tensor] = None):
    """Launch the fused fit on the current stream; returns a dict of device tensors."""
    dev = model.device
    B, K = j3d.shape[0], j3d.shape[1]
    D = 3 * (model.num_joints - 1)
    idx = _host_i32(np.asarray(list(model_joint_index)))
    if idx.shape != (K,):
        raise ValueError(f"model_joint_index has {idx.shape[0]} entries for {K} targets")
    if conf is not None:
        want = (B, K) if cfg.conf_per_frame else (K,)
        conf_p = _dev(conf, "conf", dev, want)
    else:
        conf_p = None
    out = {
        "global_orient": torch.empty((B, 3), dtype=torch.float32, device=dev),
        "body_pose": torch.empty((B, D), dtype=torch.float32, device=dev),
        "betas": torch.empty((B, model.num_betas), dtype=torch.float32, device=dev),
        "transl": torch.empty((B, 3), dtype=torch.float32, device=dev),
        "loss": torch.empty((B,), dtype=torch.float32, device=dev),
    }
    P = 3 + D + model.num_betas + 3
    if want_grad:
        out["grad"] = torch.empty((B, P), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _check(load_library().k2b_fit_world(
            model.handle, prior.handle, C.byref(cfg), B, K, _np_ptr(idx),
            _dev(j3d, "j3d", dev, (B, K, 3)), conf_p,
            _dev(global_orient, "global_orient", dev, (B, 3)), _dev(body_pose, "body_pose", dev, (B, D)),
            _dev(betas, "betas", dev, (B, model.num_betas)), _dev(transl, "transl", dev, (B, 3)),
            _dev(preserve_pose, "preserve_pose", dev, (B, D)),
            _dev(transl_prior_target, "transl_prior_target", dev, (B, 3)),
            C.c_void_p(out["global_orient"].data_ptr()), C.c_void_p(out["body_pose"].data_ptr()),
            C.c_void_p(out["betas"].data_ptr()), C.c_void_p(out["transl"].data_ptr()),
            C.c_void_p(out["loss"].data_ptr()),
            C.c_void_p(out["grad"].data_ptr()) if want_grad else None, stream), "k2b_fit_world")
    return out


def fit_world_lbfgs(model: NativeModel, prior: NativePrior, cfg: FitConfigC, model_joint_index: Sequence[int],
                    j3d: torch.Tensor, conf: Optional[torch.Tensor], global_orient: torch.Tensor, body_pose: torch.Tensor,
                    betas: torch.Tensor, transl: torch.Tensor, *, max_iter: int, lr: float,
                    preserve_pose: Optional[torch.Tensor] = None, transl_prior_target: Optional[torch.Tensor] = None,
                    want_grad: bool = False, history_size: int = 100, tolerance_grad: float = 1e-7,
                    tolerance_change: float = 1e-9):
    """The L-BFGS branch on the device (``k2b_fit_world_lbfgs``): per frame ``torch.optim.LBFGS(max_iter, lr,
    line_search_fn="strong_wolfe").step(closure)`` with this library's evaluate-only launch as the closure and the optimiser's
    state machine in a kernel of its own; only launches are queued on the current stream.  Returns the dict of ``fit_world``
    (``loss`` = the loss at the result, ``grad`` with `want_grad`)."""
    dev = model.device
    B, K = j3d.shape[0], j3d.shape[1]
    D = 3 * (model.num_joints - 1)
    idx = _host_i32(np.asarray(list(model_joint_index)))
    if idx.shape != (K,):
        raise ValueError(f"model_joint_index has {idx.shape[0]} entries for {K} targets")
    conf_p = None
    if conf is not None:
        conf_p = _dev(conf, "conf", dev, (B, K) if cfg.conf_per_frame else (K,))
    out = {
        "global_orient": torch.empty((B, 3), dtype=torch.float32, device=dev),
        "body_pose": torch.empty((B, D), dtype=torch.float32, device=dev),
        "betas": torch.empty((B, model.num_betas), dtype=torch.float32, device=dev),
        "transl": torch.empty((B, 3), dtype=torch.float32, device=dev),
        "loss": torch.empty((B,), dtype=torch.float32, device=dev),
    }
    if want_grad:
        out["grad"] = torch.empty((B, 3 + D + model.num_betas + 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _check(load_library().k2b_fit_world_lbfgs(
            model.handle, prior.handle, C.byref(cfg), B, K, _np_ptr(idx),
            _dev(j3d, "j3d", dev, (B, K, 3)), conf_p,
            _dev(global_orient, "global_orient", dev, (B, 3)), _dev(body_pose, "body_pose", dev, (B, D)),
            _dev(betas, "betas", dev, (B, model.num_betas)), _dev(transl, "transl", dev, (B, 3)),
            _dev(preserve_pose, "preserve_pose", dev, (B, D)),
            _dev(transl_prior_target, "transl_prior_target", dev, (B, 3)),
            C.c_void_p(out["global_orient"].data_ptr()), C.c_void_p(out["body_pose"].data_ptr()),
            C.c_void_p(out["betas"].data_ptr()), C.c_void_p(out["transl"].data_ptr()),
            C.c_void_p(out["loss"].data_ptr()), C.c_void_p(out["grad"].data_ptr()) if want_grad else None,
            int(max_iter), int(history_size), float(lr), float(tolerance_grad), float(tolerance_change), stream),
            "k2b_fit_world_lbfgs")
    return out


def fit_sequence_lbfgs(model: NativeModel, prior: NativePrior, cfg: FitConfigC, first_iters: int, followup_iters: int,
                       model_joint_index: Sequence[int], j3d: torch.Tensor, conf: Optional[torch.Tensor],
                       global_orient: torch.Tensor, body_pose: torch.Tensor, betas: torch.Tensor, transl: torch.Tensor, *, lr: float,
                       history_size: int = 100, tolerance_grad: float = 1e-7, tolerance_change: float = 1e-9):
    """ONE warm-start sequence under the L-BFGS branch (``k2b_fit_sequence_lbfgs``): ``j3d`` (T, K, 3), start (1, ...) of frame
    0; every later frame starts from its predecessor's result with ``cfg.pose_preserve_weight``; returns (T, ...) tensors."""
    dev = model.device
    T, K = j3d.shape[0], j3d.shape[1]
    D = 3 * (model.num_joints - 1)
    idx = _host_i32(np.asarray(list(model_joint_index)))
    if idx.shape != (K,):
        raise ValueError(f"model_joint_index has {idx.shape[0]} entries for {K} targets")
    conf_p = None
    if conf is not None:
        conf_p = _dev(conf, "conf", dev, (T, K) if cfg.conf_per_frame else (K,))
    out = {
        "global_orient": torch.empty((T, 3), dtype=torch.float32, device=dev),
        "body_pose": torch.empty((T, D), dtype=torch.float32, device=dev),
        "betas": torch.empty((T, model.num_betas), dtype=torch.float32, device=dev),
        "transl": torch.empty((T, 3), dtype=torch.float32, device=dev),
        "loss": torch.empty((T,), dtype=torch.float32, device=dev),
    }
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _check(load_library().k2b_fit_sequence_lbfgs(
            model.handle, prior.handle, C.byref(cfg), T, K, _np_ptr(idx), _dev(j3d, "j3d", dev, (T, K, 3)), conf_p,
            _dev(global_orient, "global_orient", dev, (1, 3)), _dev(body_pose, "body_pose", dev, (1, D)),
            _dev(betas, "betas", dev, (1, model.num_betas)), _dev(transl, "transl", dev, (1, 3)),
            C.c_void_p(out["global_orient"].data_ptr()), C.c_void_p(out["body_pose"].data_ptr()),
            C.c_void_p(out["betas"].data_ptr()), C.c_void_p(out["transl"].data_ptr()), C.c_void_p(out["loss"].data_ptr()),
            int(first_iters), int(followup_iters), int(history_size), float(lr), float(tolerance_grad), float(tolerance_change),
            stream), "k2b_fit_sequence_lbfgs")
    return out


def fit_sequence(model: NativeModel, prior: NativePrior, cfg: FitConfigC, followup_iters: int,
                 model_joint_index: Sequence[int], j3d: torch.Tensor, conf: Optional[torch.Tensor],
                 global_orient: torch.Tensor, body_pose: torch.Tensor, betas: torch.Tensor, transl: torch.Tensor):
    """Warm-start chains in one launch (``k2b_fit_sequence``): ``j3d`` (S, T, K, 3), start parameters (S, ...) of every
    sequence's first frame; returns (S, T, ...) tensors.  ``cfg.num_iters`` iterations for frame 0, ``followup_iters``
    for the others, ``cfg.pose_preserve_weight`` on frames >= 1."""
    dev = model.device
    S, T, K = j3d.shape[0], j3d.shape[1], j3d.shape[2]
    D = 3 * (model.num_joints - 1)
    idx = _host_i32(np.asarray(list(model_joint_index)))
    if idx.shape != (K,):
        raise ValueError(f"model_joint_index has {idx.shape[0]} entries for {K} targets")
    conf_p = None
    if conf is not None:
        conf_p = _dev(conf, "conf", dev, (S, T, K) if cfg.conf_per_frame else (K,))
    out = {
        "global_orient": torch.empty((S, T, 3), dtype=torch.float32, device=dev),
        "body_pose": torch.empty((S, T, D), dtype=torch.float32, device=dev),
        "betas": torch.empty((S, T, model.num_betas), dtype=torch.float32, device=dev),
        "transl": torch.empty((S, T, 3), dtype=torch.float32, device=dev),
        "loss": torch.empty((S, T), dtype=torch.float32, device=dev),
    }
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _check(load_library().k2b_fit_sequence(
            model.handle, prior.handle, C.byref(cfg), S, T, int(followup_iters), K, _np_ptr(idx),
            _dev(j3d, "j3d", dev, (S, T, K, 3)), conf_p,
            _dev(global_orient, "global_orient", dev, (S, 3)), _dev(body_pose, "body_pose", dev, (S, D)),
            _dev(betas, "betas", dev, (S, model.num_betas)), _dev(transl, "transl", dev, (S, 3)),
            C.c_void_p(out["global_orient"].data_ptr()), C.c_void_p(out["body_pose"].data_ptr()),
            C.c_void_p(out["betas"].data_ptr()), C.c_void_p(out["transl"].data_ptr()),
            C.c_void_p(out["loss"].data_ptr()), stream), "k2b_fit_sequence")
    return out


def angular_error_deg(pred_rotvec: torch.Tensor, gt_rotvec: torch.Tensor) -> torch.Tensor:
    """Geodesic angle in degrees between pairs of axis-angle rotations, (..., 3) x (..., 3) -> (...)
    (``k2b_angular_error_deg``; launched on the current stream of the tensors' device)."""
    dev = require_device(pred_rotvec.device if isinstance(pred_rotvec, torch.Tensor) else None)
    if tuple(pred_rotvec.shape) != tuple(gt_rotvec.shape) or pred_rotvec.shape[-1] != 3:
        raise ValueError(f"expected two (...,3) tensors of equal shape, got {tuple(pred_rotvec.shape)} and {tuple(gt_rotvec.shape)}")
    n = pred_rotvec.numel() // 3
    out = torch.empty(pred_rotvec.shape[:-1], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _check(load_library().k2b_angular_error_deg(n, _dev(pred_rotvec, "pred_rotvec", dev), _dev(gt_rotvec, "gt_rotvec", dev),
                                                    C.c_void_p(out.data_ptr()) if n else None, stream), "k2b_angular_error_deg")
    return out


def vertex_term(model: NativeModel, extra_index: Sequence[int], targets: torch.Tensor, conf: Optional[torch.Tensor],
                sigma: float, joint_loss_weight: float, global_orient: torch.Tensor, body_pose: torch.Tensor,
                betas: torch.Tensor, transl: torch.Tensor):
    """Loss (B,) and gradient (B, P) of the joint-loss term of vertex-selected joints (``k2b_vertex_term``)."""
    dev = model.device
    B, E = targets.shape[0], targets.shape[1]
    D = 3 * (model.num_joints - 1)
    idx = _host_i32(np.asarray(list(extra_index)))
    if idx.shape != (E,):
        raise ValueError(f"extra_index has {idx.shape[0]} entries for {E} targets")
    loss = torch.empty((B,), dtype=torch.float32, device=dev)
    grad = torch.empty((B, 3 + D + model.num_betas + 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _check(load_library().k2b_vertex_term(
            model.handle, B, E, _np_ptr(idx), _dev(targets, "targets", dev, (B, E, 3)), _dev(conf, "conf", dev, (E,)),
            float(sigma), float(joint_loss_weight), _dev(global_orient, "global_orient", dev, (B, 3)),
            _dev(body_pose, "body_pose", dev, (B, D)), _dev(betas, "betas", dev, (B, model.num_betas)),
            _dev(transl, "transl", dev, (B, 3)), C.c_void_p(loss.data_ptr()), C.c_void_p(grad.data_ptr()), stream),
            "k2b_vertex_term")
    return loss, grad


def adam_step(params: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, step_size: float,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
    """In-place ``torch.optim.Adam`` single-tensor step ``step`` (1-based) on a flat float32 block (``k2b_adam_step``)."""
    dev = require_device(params.device)
    n = params.numel()
    for name, t in (("grad", grad), ("m", m), ("v", v)):
        if t.numel() != n:
            raise ValueError(f"{name} has {t.numel()} elements, params {n}")
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _check(load_library().k2b_adam_step(n, _dev(params, "params", dev), _dev(grad, "grad", dev), _dev(m, "m", dev),
                                            _dev(v, "v", dev), int(step), float(step_size), float(beta1), float(beta2),
                                            float(eps), stream), "k2b_adam_step")
