"""Model factory with the reference's contract (models/get_model.py:4-13)."""
from .models import BaseModel, DefaultModel
from .packnet import PacknetModel


def get_model(cfg, id):
    if cfg.data.model_name == "default":
        return DefaultModel(cfg, id)
    if cfg.data.model_name == "base":
        return BaseModel(cfg, id)
    if cfg.data.model_name == "packnet":
        # the host object and its sweep + DPV head; the PackNet CNN itself is plugged in by the caller (models/packnet.py)
        return PacknetModel(cfg, id)
    raise NotImplementedError(cfg.data.model_name)
