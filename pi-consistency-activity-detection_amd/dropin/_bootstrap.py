"""Puts the repo root on sys.path and imports the package (hyphenated directory -> alias picons_amd)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import picons_amd  # noqa: E402,F401
