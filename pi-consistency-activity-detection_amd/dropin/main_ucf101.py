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


# (flag, type or None for a switch, default, what it does) -- names, types and defaults are the reference's CLI
# (main_ucf101.py:285-315, main_jhmdb.py:283-310); the descriptions are this project's.
_COMMON_FLAGS = [
    ("bs", int, 16, "clips per step; half of them labeled, half unlabeled"),
    ("epochs", int, 1, "passes over the unlabeled set; also the length of the consistency ramp-up"),
    ("model_name", str, "i3d", "accepted for compatibility, not used"),
    ("lr", float, 0.001, "Adam step size at the start (ReduceLROnPlateau lowers it)"),
    ("pf", int, 50, "print running means every this many steps"),
    ("exp_id", str, "debug", "sub-directory of train_log_wts/ for checkpoints"),
    ("const_loss", str, "l2", "must be one of jsd / l2 / l1; the step always uses the L2 form, as the reference does"),
    ("wt_cls", float, 1, "weight of the spread (classification) loss"),
    ("wt_cons", float, 1, "weight of the flip-consistency loss"),
    ("seed", int, 47, "seeds python / numpy / torch RNGs"),
    ("thresh_epoch", int, 11, "from this epoch on unlabeled clips are masked with their predicted class instead of all classes"),
    ("n_frames", int, 3, "temporal window (3 or 5 frames) of the variance attention mask"),
    ("bv", None, False, "weight the consistency loss with the temporal-variance attention mask"),
    ("predict_maps", None, False, "take the variance of sigmoid(logits) instead of the logits"),
    ("cyclic", None, False, "accepted for compatibility, not used"),
    ("gv", None, False, "weight the consistency loss with the second-order temporal-gradient mask"),
    ("lower_thresh", float, None, "gradient mask: probabilities below this count as 0"),
    ("upper_thresh", float, None, "gradient mask: probabilities above this count as 1"),
]


def build_parser(extra):
    parser = argparse.ArgumentParser(description="semi-supervised action detection: flip consistency with variance / gradient attention")
    for name, typ, default, text in _COMMON_FLAGS + extra:
        if typ is None:
            parser.add_argument("--" + name, action="store_true", help=text)
        else:
            parser.add_argument("--" + name, type=typ, default=default, help=text)
    return parser


def parse_args(argv=None):
    """The flag set of main_ucf101.py:285-315."""
    return build_parser([
        ("pretrained", str, "i3d", "accepted for compatibility, not used"),
        ("loc_loss", str, "dice", "localisation loss next to BCE; only 'dice' exists"),
        ("pkl_file_label", str, "train_annots_20_labeled.pkl", "annotation file of the labeled split"),
        ("pkl_file_unlabel", str, "train_annots_80_unlabeled.pkl", "annotation file of the unlabeled split"),
        ("wt_loc", float, 1, "weight of the localisation (BCE + Dice) loss"),
        ("workers", int, 8, "DataLoader worker processes"),
        ("bv_wt", float, 0.5, "share of the variance-mask term when --bv and --gv are both given"),
        ("gv_wt", float, 0.5, "share of the gradient-mask term when --bv and --gv are both given"),
    ]).parse_args(argv)


def run(args):
    global model, criterion_cls, criterion_seg_1, criterion_seg_2
    if args.loc_loss != 'dice':
        print("--loc_loss %r is not available (only 'dice'); stopping" % args.loc_loss)      # 'iou' is a NameError in the reference (:396)
        sys.exit(1)
    if args.const_loss not in ('jsd', 'l2', 'l1'):
        print("--const_loss %r is not one of jsd / l2 / l1; stopping" % args.const_loss)
        sys.exit(1)
    rank, world, local = pdist.init_from_env()
    torch.cuda.set_device(local)
    if args.seed:
        random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed + rank)
    hw = int(os.environ.get("PICONS_HW", "224"))
    fused = os.environ.get("PICONS_FUSED", "1") != "0"
    steps = int(os.environ.get("PICONS_STEPS", "4"))
    mode = os.environ.get("PICONS_SYNTHETIC", "1")          # read only: the process environment is not written to
    if mode not in ("1", "u8"):
        raise RuntimeError("real UCF101/JHMDB loaders need skvideo/cv2 + the dataset, neither is available here; set PICONS_SYNTHETIC=1 (or u8)")
    n = args.bs // 2
    Loader = SyntheticVideoLoader if (mode == "u8" and hw == 224 and DATASET == "ucf101") else SyntheticLoader
    labeled_loader = Loader(n, True, steps, rank, NUM_CLASSES, hw, 0)
    unlabeled_loader = Loader(n, False, steps, rank, NUM_CLASSES, hw, 1)
    val_loader = SyntheticLoader(args.bs, True, 1, rank, NUM_CLASSES, hw, 5)
    print(len(labeled_loader), len(unlabeled_loader), len(val_loader))

    # the reference's CapsNet() loads ../weights/rgb_charades.pt and raises without it (capsules_ucf101.py:344).  In synthetic mode a
    # missing file is expected (no network for the download): pt_path=None is passed EXPLICITLY, so a caller who supplies the weights, or
    # constructs CapsNet() himself, still gets the hard error for a wrong path
    pt_path = '../weights/rgb_charades.pt'
    if not os.path.exists(pt_path):
        print("WARNING: %s not found; synthetic mode (PICONS_SYNTHETIC=%s): the I3D trunk keeps its random initialisation" % (pt_path, mode), file=sys.stderr)
        pt_path = None
    model = CapsNet(pt_path=pt_path, num_classes=NUM_CLASSES, hw=hw, seed=args.seed) if NUM_CLASSES != 24 or hw != 224 else CapsNet(pt_path=pt_path, seed=args.seed)
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
            print("validation loss improved: %.4f -> %.4f, checkpoint written" % (prev_best_val_loss, val_loss))
            p = os.path.join(model_save_dir, f'best_model_val_loss_{e}.pth')
            torch.save(model.state_dict(), p)
            prev_best_val_loss = val_loss
            if prev_val_path and e < 20:
                os.remove(prev_val_path)
            prev_val_path = p
        if rank == 0 and train_loss < prev_best_train_loss:
            print("training loss improved: %.4f -> %.4f, checkpoint written" % (prev_best_train_loss, train_loss))
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
