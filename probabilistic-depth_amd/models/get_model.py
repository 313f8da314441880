"""Model factory with the reference's contract (models/get_model.py:4-13)."""
from .models import BaseModel, DefaultModel


def get_model(cfg, id):
    if cfg.data.model_name == "default":
        return DefaultModel(cfg, id)
    if cfg.data.model_name == "base":
        return BaseModel(cfg, id)
    if cfg.data.model_name == "packnet":
        # The PackNet encoder/decoder is a separate dense CNN that is out of this package's scope
        # (SURVEY.md section 2 row 5); its hot path -- sweep -> log_softmax -> E[d], packnet.py:343-396 -- is
        # models.packnet_head.PacknetHead (one fused launch for the whole batch).
        raise NotImplementedError("packnet: network not provided; its sweep+DPV head is models.packnet_head.PacknetHead")
    raise NotImplementedError(cfg.data.model_name)
