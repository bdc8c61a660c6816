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
    parser = argparse.ArgumentParser(description='loc var const')
    parser.add_argument('--bs', type=int, default=16, help='mini-batch size')
    parser.add_argument('--pf', type=int, default=50, help='print frequency every batch')
    parser.add_argument('--epochs', type=int, default=1, help='number of total epochs to run')
    parser.add_argument('--model_name', type=str, default='i3d', help='model name')
    parser.add_argument('--lr', type=float, default=0.001, help='learning rate')
    parser.add_argument('--seg_loss', type=str, default='dice', help='dice or iou loss')
    parser.add_argument('--exp_id', type=str, default='debug', help='experiment name')
    parser.add_argument('--pkl_file_label', type=str, default='jhmdb_classes_list_per_20_labeled.txt', help='label subset')
    parser.add_argument('--pkl_file_unlabel', type=str, default='jhmdb_classes_list_per_80_unlabeled.txt', help='unlabele subset')
    parser.add_argument('--const_loss', type=str, default="l2", help='consistency loss type')
    parser.add_argument('--wt_seg', type=float, default=1, help='segmentation loss weight')
    parser.add_argument('--wt_cls', type=float, default=1, help='Classification loss weight')
    parser.add_argument('--wt_cons', type=float, default=1, help='class consistency loss weight')
    parser.add_argument('--seed', type=int, default=47, help='seed for initializing training.')
    parser.add_argument('--thresh_epoch', type=int, default=11, help='thresh epoch to introduce pseudo labels')
    parser.add_argument('--n_frames', type=int, default=3, help='batch variance frames number.')
    parser.add_argument('--bv', action='store_true', help='use batch variance')
    parser.add_argument('--predict_maps', action='store_true', help='use sigmoid outputs')
    parser.add_argument('--cyclic', action='store_true', help='use batch variance')
    parser.add_argument('--gv', action='store_true', help='use grad variance')
    parser.add_argument('--lower_thresh', type=float, default=None, help='lower conf thresh')
    parser.add_argument('--upper_thresh', type=float, default=None, help='upper conf thresh')
    parser.add_argument('--viz', action='store_true', help='map visuzlization debug')
    parser.add_argument('--seed_num', type=int, default=47, help='seed variation pickle files')
    a = parser.parse_args(argv)
    a.wt_loc, a.loc_loss, a.bv_wt, a.gv_wt, a.workers = a.wt_seg, a.seg_loss, 0.5, 0.5, 8
    return a


if __name__ == '__main__':
    M.NUM_CLASSES, M.DATASET = 21, "jhmdb"
    a = parse_args()
    print(vars(a))
    M.run(a)
