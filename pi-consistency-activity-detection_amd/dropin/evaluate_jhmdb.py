#!/usr/bin/env python3
"""Drop-in for /root/reference/evaluate_jhmdb.py: the loop of evaluate_ucf101.py with the 21-class head (:45)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["PICONS_DATASET"] = "jhmdb"
from evaluate_ucf101 import iou  # noqa: E402

if __name__ == '__main__':
    iou('train')
