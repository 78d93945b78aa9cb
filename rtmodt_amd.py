"""Importable alias for the package directory
``real-time-multi-object-detection---tracking-system_amd/`` (whose name is not a
Python identifier).  ``import rtmodt_amd`` returns that package object."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

_pkg = importlib.import_module("real-time-multi-object-detection---tracking-system_amd")
sys.modules[__name__] = _pkg
