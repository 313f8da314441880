"""Importable alias for the package directory ``probabilistic-depth_amd`` (hyphenated name)."""
import importlib
import sys

_pkg = importlib.import_module("probabilistic-depth_amd")
sys.modules[__name__] = _pkg
