"""Drop-in for /root/reference/models/pytorch_i3d.py: `from models.pytorch_i3d import InceptionI3d`."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _bootstrap  # noqa: E402,F401
from picons_amd.model import InceptionI3d  # noqa: E402,F401
