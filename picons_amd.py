"""Importable alias for the hyphen-named package directory `pi-consistency-activity-detection_amd/`.

`import picons_amd` returns that package (its __init__ registers itself and every submodule
under both names, so there is exactly one instance of each module).
"""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("pi-consistency-activity-detection_amd")
sys.modules[__name__] = _pkg
