"""Body-model loading (reference ``keypoints2body/api/model_factory.py:19-40``).

The reference calls ``smplx.create(model_dir, model_type, gender, ext, batch_size)``.  Here a
model is a ``BodyModel`` whose constants live in HBM; it is built from, in order:
``<model_dir>/<model_type>_<gender>.npz`` (this engine's own container: the arrays smplx
exposes), or, when the ``smplx`` package and the licensed files are installed, from
``smplx.create`` (its constants are copied to the device, its Python forward is not used).
"""
from __future__ import annotations

from pathlib import Path

from ..core.config import BodyModelConfig, ModelType
from ..models.body_model import BodyModel

MODEL_EXT_DEFAULTS: dict[ModelType, str] = {"smpl": "pkl", "smplh": "pkl", "smplx": "npz", "mano": "pkl", "flame": "pkl"}


def load_body_model(config: BodyModelConfig, device=None) -> BodyModel:
    model_dir = Path(config.model_dir).expanduser()
    own = model_dir / f"{config.model_type}_{config.gender}.npz"
    if own.exists():
        return BodyModel.from_npz(str(own), device=device)
    try:
        import smplx  # optional: only to read the licensed model files
    except ImportError as e:
        raise FileNotFoundError(
            f"no body model found: {own} does not exist and the 'smplx' package is not installed to read "
            f"{model_dir}/{config.model_type}/. Pass model=BodyModel(...) or convert the model to {own.name}") from e
    ext = config.ext or MODEL_EXT_DEFAULTS[config.model_type]
    ref = smplx.create(str(model_dir), model_type=config.model_type, gender=config.gender, ext=ext,
                       batch_size=config.batch_size)
    return BodyModel.from_smplx(ref, device=device)
