"""Drop-in for /root/reference/utils/ramp_ups.py (only exp_rampup is on the hot path, main_ucf101.py:419)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _bootstrap  # noqa: E402,F401
from picons_amd.step import exp_rampup  # noqa: E402,F401
