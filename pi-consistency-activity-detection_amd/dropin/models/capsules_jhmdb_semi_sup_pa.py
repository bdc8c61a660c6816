"""The 21-class model file main_jhmdb.py:369 imports is ABSENT from the reference; this is
the same CapsNet with ConvCaps(32, 21) / upsample1(21*16 -> 64), inferred from main_jhmdb.py:383 and
evaluate_jhmdb.py:45."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _bootstrap  # noqa: E402,F401
from picons_amd.model import CapsNet as _CapsNet  # noqa: E402


class CapsNet(_CapsNet):
    def __init__(self, pt_path='../weights/rgb_charades.pt', P=4, pretrained_load='i3d', **kw):
        kw.setdefault("num_classes", 21)
        super().__init__(pt_path, P, pretrained_load, **kw)
