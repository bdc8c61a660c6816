#!/usr/bin/env python3
"""Drop-in for /root/reference/main_jhmdb.py: its CLI (:283-310, `--wt_seg`/`--seg_loss`, no
--workers/--bv_wt/--gv_wt/--pretrained/--loc_loss), 21 classes, labels synthesised as ones/zeros
(:68-70), gv overriding bv (:121,132).  Everything else is shared with main_ucf101.py."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import main_ucf101 as M  # noqa: E402


def parse_args(argv=None):
    """The flag set of main_jhmdb.py:283-310 (no --workers / --bv_wt / --gv_wt / --pretrained / --loc_loss)."""
    a = M.build_parser([
        ("seg_loss", str, "dice", "localisation loss next to BCE; only 'dice' exists"),
        ("pkl_file_label", str, "jhmdb_classes_list_per_20_labeled.txt", "video list of the labeled split"),
        ("pkl_file_unlabel", str, "jhmdb_classes_list_per_80_unlabeled.txt", "video list of the unlabeled split"),
        ("wt_seg", float, 1, "weight of the localisation (BCE + Dice) loss"),
        ("viz", None, False, "accepted for compatibility, not used"),
        ("seed_num", int, 47, "accepted for compatibility, not used"),
    ]).parse_args(argv)
    a.wt_loc, a.loc_loss, a.bv_wt, a.gv_wt, a.workers = a.wt_seg, a.seg_loss, 0.5, 0.5, 8
    return a


if __name__ == '__main__':
    M.NUM_CLASSES, M.DATASET = 21, "jhmdb"
    a = parse_args()
    print(vars(a))
    M.run(a)
