"""Drop-in for /root/reference/models/capsules_ucf101.py: `from models.capsules_ucf101 import CapsNet`.
The class keeps the reference's constructor / forward / load_previous_weights / state_dict surface and
runs on the HIP kernels of libpicons.so (GPU only)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _bootstrap  # noqa: E402,F401
from picons_amd.model import CapsNet  # noqa: E402,F401
