#!/usr/bin/env python3
"""Drop-in for /root/reference/main_ucf101.py: same CLI flags (:285-315), same train/validate loop
structure and checkpoint policy (:434-456), same `train_model_interface` contract (:50-150) - with the
model, the bv/gv masks and the optimiser step running on MI355X HIP kernels.

New behaviour is switched by environment variables only, so the CLI is unchanged:
  PICONS_SYNTHETIC=1     synthetic UCF101-24-shaped minibatches (no dataset / decoder libs on the box)
  PICONS_SYNTHETIC=u8    synthetic decoded uint8 videos + box annotations through the device input pipeline (picons_amd.inputpipe)
  PICONS_STEPS=<n>       steps per epoch in synthetic mode (default 4)
  PICONS_FUSED=0         use the nn.Module + autograd path (model called twice, torch.optim.Adam) instead of
                         the fused step engine (default 1: picons_amd.step.StepEngine, both passes batched)
  PICONS_HW=<px>         frame size (default 224, the only size the reference accepts)
  RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*   data parallel over the GPUs of one node (torchrun)
"""
import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _bootstrap  # noqa: E402,F401

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from torch import optim  # noqa: E402

from models.capsules_ucf101 import CapsNet  # noqa: E402
from utils.losses import SpreadLoss, DiceLoss, weighted_mse_loss  # noqa: E402
from utils.metrics import get_accuracy, IOU2  # noqa: E402
from utils.helpers import measure_pixelwise_var_v2, measure_pixelwise_gradient  # noqa: E402
from utils import ramp_ups  # noqa: E402
from picons_amd import dist as pdist, step as pstep, synthetic  # noqa: E402

NUM_CLASSES = 24
DATASET = "ucf101"
model = criterion_cls = criterion_seg_1 = criterion_seg_2 = None


def _dev(t, dtype=None):
    t = t if torch.is_tensor(t) else torch.as_tensor(np.asarray(t))
    return t.cuda().to(dtype) if dtype is not None else t.cuda()


def val_model_interface(minibatch):
    """main_ucf101.py:33-47."""
    data = _dev(minibatch['data'], torch.float32)
    action = _dev(minibatch['action'])
    segmentation = _dev(minibatch['loc_msk'], torch.float32)
    empty_vector = torch.zeros(action.shape[0]).cuda()
    output, predicted_action, _ = model(data, action, empty_vector, 0, 0)
    class_loss, abs_class_loss = criterion_cls(predicted_action, action)
    loss1 = criterion_seg_1(output, segmentation)
    loss2 = criterion_seg_2(output, segmentation)
    loc_loss = loss1 + loss2
    total_loss = loc_loss + class_loss
    return (output, predicted_action, segmentation, action, total_loss, loc_loss, class_loss)


def train_model_interface(args, label_minibatch, unlabel_minibatch, epoch, wt_ramp):
    """main_ucf101.py:50-150 on the nn.Module path (two forward calls, autograd)."""
    cat = lambda k, dt=None: torch.cat([_dev(label_minibatch[k], dt), _dev(unlabel_minibatch[k], dt)], dim=0)
    concat_data = cat('data', torch.float32)
    concat_fl_data = cat('aug_data', torch.float32)
    concat_action = cat('action')
    concat_seg = cat('loc_msk', torch.float32)
    if DATASET == "jhmdb":            # main_jhmdb.py:68-70
        concat_labels = torch.cat([torch.ones(len(label_minibatch['action'])), torch.zeros(len(unlabel_minibatch['action']))]).cuda()
    else:
        concat_labels = cat('label_vid')
    random_indices = torch.randperm(len(concat_labels)).cuda()
    concat_data, concat_fl_data = concat_data[random_indices], concat_fl_data[random_indices]
    concat_action, concat_labels, concat_seg = concat_action[random_indices], concat_labels[random_indices], concat_seg[random_indices]
    labeled_vid_index = torch.where(concat_labels == 1)[0]

    output, predicted_action, feat = model(concat_data, concat_action, concat_labels, epoch, args.thresh_epoch)
    flip_op, _, _ = model(concat_fl_data, concat_action, concat_labels, epoch, args.thresh_epoch)

    labeled_op = output[labeled_vid_index]
    labeled_seg_data = concat_seg[labeled_vid_index]
    loc_loss = criterion_seg_1(labeled_op, labeled_seg_data) + criterion_seg_2(labeled_op, labeled_seg_data)
    class_loss, abs_class_loss = criterion_cls(predicted_action[labeled_vid_index], concat_action[labeled_vid_index])

    flipped_pred_seg_map = torch.flip(flip_op, [4])
    loss_wt_simple_l2 = weighted_mse_loss(flipped_pred_seg_map, output, torch.ones_like(output))
    total_seg_cons_loss_1 = total_seg_cons_loss_2 = None
    if args.bv:
        v_c = measure_pixelwise_var_v2(output, torch.flip(flipped_pred_seg_map, [2]), frames_cnt=args.n_frames, use_sig_output=args.predict_maps)
        v_a = measure_pixelwise_var_v2(torch.flip(output, [2]), flipped_pred_seg_map, frames_cnt=args.n_frames, use_sig_output=args.predict_maps)
        loss_wt_var_1 = weighted_mse_loss(flipped_pred_seg_map, output, v_c)
        loss_wt_var_2 = weighted_mse_loss(flipped_pred_seg_map, output, torch.flip(v_a, [2]))
        total_seg_cons_loss_1 = (wt_ramp * (loss_wt_var_1 + loss_wt_var_2)) + ((1 - wt_ramp) * loss_wt_simple_l2)
    if args.gv:
        batch_grad = measure_pixelwise_gradient(output, conf_thresh_lower=args.lower_thresh, conf_thresh_upper=args.upper_thresh)
        total_seg_cons_loss_2 = weighted_mse_loss(flipped_pred_seg_map, output, batch_grad)
    if DATASET == "jhmdb":
        total_cons_loss = total_seg_cons_loss_2 if args.gv else (total_seg_cons_loss_1 if args.bv else loss_wt_simple_l2)
    elif args.bv and args.gv:
        total_cons_loss = args.bv_wt * total_seg_cons_loss_1 + args.gv_wt * total_seg_cons_loss_2
    elif args.gv:
        total_cons_loss = total_seg_cons_loss_2
    elif args.bv:
        total_cons_loss = total_seg_cons_loss_1
    else:
        total_cons_loss = loss_wt_simple_l2
    total_loss = args.wt_loc * loc_loss + args.wt_cls * class_loss + args.wt_cons * total_cons_loss
    return (output, predicted_action, concat_seg, concat_action, total_loss, loc_loss, class_loss, total_cons_loss)


class SyntheticLoader:
    """Stands in for torch DataLoader(UCF101DataLoader(...)) with the same minibatch dict contract."""

    def __init__(self, n, labeled, steps, rank, num_classes, hw, salt):
        self.n, self.labeled, self.steps, self.rank, self.nc, self.hw, self.salt = n, labeled, steps, rank, num_classes, hw, salt

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            mb = synthetic.make_minibatch(self.n, self.labeled, (1234 + self.rank) * 7919 + 2 * i + self.salt, self.nc, self.hw)
            yield {k: torch.from_numpy(v) for k, v in mb.items()}


class SyntheticVideoLoader(SyntheticLoader):
    """PICONS_SYNTHETIC=u8: synthetic DECODED videos (uint8 frames + box annotations, what `load_video` has after `vread`) go
    through the device input pipeline (picons_amd.inputpipe.get_item = UCF101DataLoader.__getitem__ from that point on) and are
    collated like torch's default collate does, as device fp32 tensors."""

    def __iter__(self):
        from picons_amd import inputpipe
        for i in range(self.steps):
            samples = []
            for j in range(self.n):
                frames, ann = synthetic.make_decoded_video(((1234 + self.rank) * 7919 + 2 * i + self.salt) * 64 + j, self.labeled, self.nc)
                samples.append(inputpipe.get_item(frames, ann, train=True))
            yield {'data': torch.stack([s['data'] for s in samples]), 'aug_data': torch.stack([s['aug_data'] for s in samples]),
                   'loc_msk': torch.stack([s['loc_msk'] for s in samples]), 'action': torch.stack([s['action'] for s in samples]),
                   'label_vid': torch.tensor([s['label_vid'] for s in samples])}


def train(args, model, labeled_train_loader, unlabeled_train_loader, optimizer, epoch, save_path, writer, ramp_wt, engine=None, reducer=None):
    """main_ucf101.py:155-223."""
    model.train(mode=True)
    model.training = True
    total_loss, accuracy, loc_loss, class_loss, class_consistency_loss = [], [], [], [], []
    steps = len(unlabeled_train_loader)
    start_time = time.time()
    labeled_iterloader = iter(labeled_train_loader)
    for batch_id, unlabel_minibatch in enumerate(unlabeled_train_loader):
        try:
            label_minibatch = next(labeled_iterloader)
        except StopIteration:
            labeled_iterloader = iter(labeled_train_loader)
            label_minibatch = next(labeled_iterloader)
        if engine is not None:            # fused HIP step: both passes batched, losses + Adam on device
            bs = len(label_minibatch['action']) + len(unlabel_minibatch['action'])
            perm = torch.randperm(bs).numpy()
            drops = [(torch.rand(bs, c) < 0.5).float().numpy() * 2 for c in (832, 128, 832, 128)]
            s = engine.train_step(label_minibatch, unlabel_minibatch, epoch, ramp_wt(epoch), perm, drops, lr=optimizer.param_groups[0]['lr'], reducer=reducer)
            _out, _flip, pred = engine.outputs()
            total_loss.append(s['total']); loc_loss.append(s['loc']); class_loss.append(s['cls']); class_consistency_loss.append(s['cons'])
            accuracy.append(get_accuracy(pred, engine.action_host))
        else:
            optimizer.zero_grad()
            output, predicted_action, segmentation, action, loss, s_loss, c_loss, cc_loss = \
                train_model_interface(args, label_minibatch, unlabel_minibatch, epoch, ramp_wt(epoch))
            loss.backward()
            optimizer.step()
            total_loss.append(loss.item()); loc_loss.append(s_loss.item()); class_loss.append(c_loss.item())
            class_consistency_loss.append(cc_loss.item()); accuracy.append(get_accuracy(predicted_action, action))
        if (batch_id + 1) % args.pf == 0:
            print(f'[TRAIN] epoch-{epoch:0{len(str(args.epochs))}}/{args.epochs}, batch-{batch_id+1:0{len(str(steps))}}/{steps},'
                  f'loss-{np.mean(total_loss):.3f}, acc-{np.mean(accuracy):.3f}'
                  f'\t [LOSS ] cls-{np.mean(class_loss):.3f}, seg-{np.mean(loc_loss):.3f}, const-{np.mean(class_consistency_loss):.3f}')
            sys.stdout.flush()
    print("Training time: ", time.time() - start_time)
    return float(np.array(total_loss).mean())


def validate(model, val_data_loader, epoch):
    """main_ucf101.py:226-278."""
    model.eval()
    model.training = False
    total_loss, accuracy, total_IOU, validiou = [], [], 0, 0
    with torch.no_grad():
        for minibatch in val_data_loader:
            output, predicted_action, segmentation, action, loss, s_loss, c_loss = val_model_interface(minibatch)
            total_loss.append(loss.item())
            accuracy.append(get_accuracy(predicted_action, action))
            maskout_np = (output.cpu().numpy() > 0).astype(np.float32)
            truth_np = segmentation.cpu().numpy()
            for a in range(maskout_np.shape[0]):
                iou = IOU2(truth_np[a], maskout_np[a])
                if iou == iou:
                    total_IOU += iou; validiou += 1
    print(f'[VAL] epoch-{epoch}, loss-{np.mean(total_loss):.3f}, acc-{np.mean(accuracy):.3f} [IOU ] {total_IOU / max(validiou, 1):.3f}')
    return float(np.mean(total_loss))


def parse_args(argv=None):
    """Flag names, types and defaults of main_ucf101.py:285-315."""
    parser = argparse.ArgumentParser(description='loc var const')
    parser.add_argument('--bs', type=int, default=16, help='mini-batch size')
    parser.add_argument('--epochs', type=int, default=1, help='number of total epochs to run')
    parser.add_argument('--model_name', type=str, default='i3d', help='model name')
    parser.add_argument('--lr', type=float, default=0.001, help='learning rate')
    parser.add_argument('--pf', type=int, default=50, help='print frequency every batch')
    parser.add_argument('--pretrained', type=str, default="i3d", help='loading pretrained model')
    parser.add_argument('--loc_loss', type=str, default='dice', help='dice or iou loss')
    parser.add_argument('--exp_id', type=str, default='debug', help='experiment name')
    parser.add_argument('--pkl_file_label', type=str, default='train_annots_20_labeled.pkl', help='label subset')
    parser.add_argument('--pkl_file_unlabel', type=str, default='train_annots_80_unlabeled.pkl', help='unlabele subset')
    parser.add_argument('--const_loss', type=str, default='l2', help='consistency loss type')
    parser.add_argument('--wt_loc', type=float, default=1, help='segmentation loss weight')
    parser.add_argument('--wt_cls', type=float, default=1, help='Classification loss weight')
    parser.add_argument('--wt_cons', type=float, default=1, help='class consistency loss weight')
    parser.add_argument('--seed', type=int, default=47, help='seed for initializing training.')
    parser.add_argument('--thresh_epoch', type=int, default=11, help='thresh epoch to introduce pseudo labels')
    parser.add_argument('--workers', type=int, default=8, help='num workers')
    parser.add_argument('--n_frames', type=int, default=3, help='batch variance frames number.')
    parser.add_argument('--bv', action='store_true', help='use batch variance')
    parser.add_argument('--predict_maps', action='store_true', help='use sigmoid outputs')
    parser.add_argument('--bv_wt', type=float, default=0.5, help='batch variance weight')
    parser.add_argument('--cyclic', action='store_true', help='use batch variance')
    parser.add_argument('--gv', action='store_true', help='use grad variance')
    parser.add_argument('--lower_thresh', type=float, default=None, help='lower conf thresh')
    parser.add_argument('--upper_thresh', type=float, default=None, help='upper conf thresh')
    parser.add_argument('--gv_wt', type=float, default=0.5, help='grad variance weight')
    return parser.parse_args(argv)


def run(args):
    global model, criterion_cls, criterion_seg_1, criterion_seg_2
    if args.loc_loss != 'dice':
        print("wrong parameter recheck. Exiting the code !!!!")      # 'iou' is a NameError in the reference (:396)
        sys.exit(1)
    if args.const_loss not in ('jsd', 'l2', 'l1'):
        print("no consistency criterion found. Exiting the code!!!")
        sys.exit(1)
    rank, world, local = pdist.init_from_env()
    torch.cuda.set_device(local)
    if args.seed:
        random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed + rank)
    hw = int(os.environ.get("PICONS_HW", "224"))
    fused = os.environ.get("PICONS_FUSED", "1") != "0"
    steps = int(os.environ.get("PICONS_STEPS", "4"))
    mode = os.environ.setdefault("PICONS_SYNTHETIC", "1")
    if mode not in ("1", "u8"):
        raise RuntimeError("real UCF101/JHMDB loaders need skvideo/cv2 + the dataset, neither is available here; set PICONS_SYNTHETIC=1 (or u8)")
    n = args.bs // 2
    Loader = SyntheticVideoLoader if (mode == "u8" and hw == 224 and DATASET == "ucf101") else SyntheticLoader
    labeled_loader = Loader(n, True, steps, rank, NUM_CLASSES, hw, 0)
    unlabeled_loader = Loader(n, False, steps, rank, NUM_CLASSES, hw, 1)
    val_loader = SyntheticLoader(args.bs, True, 1, rank, NUM_CLASSES, hw, 5)
    print(len(labeled_loader), len(unlabeled_loader), len(val_loader))

    model = CapsNet(num_classes=NUM_CLASSES, hw=hw, seed=args.seed) if NUM_CLASSES != 24 or hw != 224 else CapsNet(seed=args.seed)
    model = model.cuda()
    criterion_cls = SpreadLoss(num_class=NUM_CLASSES, m_min=0.2, m_max=0.9)
    criterion_seg_1 = nn.BCEWithLogitsLoss()
    criterion_seg_2 = DiceLoss()
    optimizer = optim.Adam(model.parameters(), lr=args.lr, weight_decay=0, eps=1e-6)
    scheduler = optim.lr_scheduler.ReduceLROnPlateau(optimizer, 'min', min_lr=1e-7, patience=5, factor=0.1)
    ramp_wt = ramp_ups.exp_rampup(args.epochs)
    engine = reducer = None
    if world > 1 and not fused:
        # the nn.Module path has no gradient exchange: N ranks would train N independent models and rank 0's would be saved
        raise RuntimeError("WORLD_SIZE > 1 needs the fused step (PICONS_FUSED=1): the autograd path does no gradient all-reduce")
    if fused:
        engine = pstep.StepEngine(args, bs=args.bs, hw=hw, num_classes=NUM_CLASSES, jhmdb=(DATASET == "jhmdb"), state=model.state_dict(),
                                  device="cuda:%d" % local)
        reducer = engine.make_reducer() if world > 1 else None
    save_path = os.path.join('train_log_wts', args.exp_id)
    model_save_dir = os.path.join(save_path, time.strftime('%m-%d-%H-%M'))
    os.makedirs(model_save_dir, exist_ok=True)
    prev_best_val_loss = prev_best_train_loss = 10000
    prev_val_path = prev_train_path = None
    for e in range(1, args.epochs + 1):
        train_loss = train(args, model, labeled_loader, unlabeled_loader, optimizer, e, save_path, None, ramp_wt, engine, reducer)
        if engine is not None:
            model.load_state_dict(engine.state_dict())
        val_loss = validate(model, val_loader, e)
        # data parallel: every rank must take the same scheduler / checkpoint decisions, so they are taken on the mean of the
        # per-rank losses (each rank saw its own shard); otherwise ReduceLROnPlateau fires on one rank and the replicas
        # train at different learning rates from then on
        train_loss, val_loss = pdist.mean_over_ranks([train_loss, val_loss], device="cuda:%d" % local)
        if rank == 0 and val_loss < prev_best_val_loss:           # checkpoint policy of main_ucf101.py:439-455
            print("Yay!!! Got the val loss down...")
            p = os.path.join(model_save_dir, f'best_model_val_loss_{e}.pth')
            torch.save(model.state_dict(), p)
            prev_best_val_loss = val_loss
            if prev_val_path and e < 20:
                os.remove(prev_val_path)
            prev_val_path = p
        if rank == 0 and train_loss < prev_best_train_loss:
            print("Yay!!! Got the train loss down...")
            p = os.path.join(model_save_dir, f'best_model_train_loss_{e}.pth')
            torch.save(model.state_dict(), p)
            prev_best_train_loss = train_loss
            if prev_train_path and e < 20:
                os.remove(prev_train_path)
            prev_train_path = p
        scheduler.step(train_loss)
    return train_loss


if __name__ == '__main__':
    a = parse_args()
    print(vars(a))
    run(a)
