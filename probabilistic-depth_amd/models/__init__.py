from .get_model import get_model  # noqa: F401
