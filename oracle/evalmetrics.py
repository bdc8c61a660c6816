"""CPU oracle (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py): the reference's evaluation arithmetic, restated in
numpy from /root/reference/evaluate_ucf101.py (evaluate_jhmdb.py is the same loop with 21 classes).

Pinned: tests/golden/eval_map.npz holds the accumulators of the reference's own loop run on the synthetic videos of
tests/evalfixture.py (tools/make_eval_golden.py); tests/test_oracle_golden.py checks this restatement against them."""
import numpy as np
import torch

N_THR = 20
# `i_over_u >= iou_threshs[k]` (evaluate_ucf101.py:168,176) compares a Python float with an np.float32 SCALAR.  Under the
# numpy the reference pins (1.21, requirements.txt:13) two scalars promote to float64, so the ratio is compared with the
# float32 threshold widened to float64; numpy >= 2 (NEP 50, this container) would round the ratio to float32 instead.  The
# two differ only when a ratio equals k/20 exactly (1/20 vs float32(0.05) = 0.05000000075).  The restatement and the
# HIP kernel follow the pinned environment (float64); the golden run contains no such tie, so it is the same under both.


def make_clips(video, bbox, label, f_skip=2):
    """evaluate_ucf101.py:79-97: windows of 16 frames, both phases of the frame skip, 8 frames each at stride f_skip, zero
    frames past the end; clips whose truth is empty are dropped.  video [F,H,W,3], bbox [F,H,W,1] -> [(clip, boxes, label)]."""
    video, bbox = np.asarray(video), np.asarray(bbox)
    F = video.shape[0]
    clips = []
    for i in range(0, F, 8 * f_skip):
        for j in range(f_skip):
            v = np.zeros((8,) + video.shape[1:], np.float32)
            b = np.zeros((8,) + bbox.shape[1:], np.float32)
            for k in range(8):
                ind = i + j + k * f_skip
                if ind < F:
                    v[k] = video[ind]; b[k] = bbox[ind]
            if np.sum(b) != 0:
                clips.append((v, b, label))
    return clips


class MapState:
    """The accumulators of evaluate_ucf101.py:66-72."""

    def __init__(self, n_classes=24):
        self.n_classes = n_classes
        self.n_correct = 0
        self.n_vids = np.zeros((n_classes, 1))
        self.n_tot_frames = np.zeros((n_classes, 1))
        self.frame_ious = np.zeros((n_classes, N_THR))
        self.video_ious = np.zeros((n_classes, N_THR))
        self.iou_threshs = np.arange(0, N_THR, dtype=np.float32) / 20

    def add_video(self, seg_logits, gt, predictions, label):
        """evaluate_ucf101.py:128-183.  seg_logits (B,1,8,H,W) network output, gt (B*8,H,W,1) truth frames in the same order,
        predictions (B, n_classes)."""
        label = int(label)
        seg = torch.sigmoid(torch.as_tensor(seg_logits, dtype=torch.float32)).numpy()          # :128
        seg = np.transpose(seg, [0, 2, 3, 4, 1]).reshape((-1,) + gt.shape[1:])                  # :130,:146
        fin_pred = int(np.argmax(np.mean(np.asarray(predictions), axis=0)))                     # :140-142
        if fin_pred == label:
            self.n_correct += 1
        pred = (seg >= 0.5).astype(np.int64)                                                   # :148
        seg_plus_gt = pred + gt                                                                # :149
        vid_inter, vid_union = 0, 0
        for i in range(gt.shape[0]):                                                           # :156-171
            if np.sum(gt[i]) == 0:
                continue
            self.n_tot_frames[label] += 1
            inter = np.count_nonzero(seg_plus_gt[i] == 2)
            union = np.count_nonzero(seg_plus_gt[i])
            vid_inter += inter; vid_union += union
            i_over_u = inter / union
            self.frame_ious[label] += np.float64(i_over_u) >= self.iou_threshs.astype(np.float64)   # see _ge
        self.n_vids[label] += 1                                                                # :173-177
        self.video_ious[label] += np.float64(vid_inter / vid_union) >= self.iou_threshs.astype(np.float64)

    def result(self):
        """evaluate_ucf101.py:181-186 (classes without a video give NaN, as there)."""
        with np.errstate(invalid="ignore", divide="ignore"):
            fAP = self.frame_ious / self.n_tot_frames
            vAP = self.video_ious / self.n_vids
            return dict(accuracy=self.n_correct / np.sum(self.n_vids), fmAP=np.mean(fAP, axis=0), vmAP=np.mean(vAP, axis=0))


def evaluate(model, videos, n_classes=24, clip_batch_size=14):
    """evaluate_ucf101.py:73-186 for one checkpoint: clips -> batches of 14 -> model(data, 500s, 500s, 0, 0) -> accumulate."""
    st = MapState(n_classes)
    for video, bbox, label in videos:
        clips = make_clips(video, bbox, label)
        if not clips:
            continue                                                                           # :99-101
        segs, preds, gts = [], [], []
        for i in range(0, len(clips), clip_batch_size):
            batch = clips[i:i + clip_batch_size]
            data = torch.from_numpy(np.transpose(np.stack([c[0] for c in batch]), [0, 4, 1, 2, 3])).float()
            empty = torch.full((len(batch), 1), 500, dtype=torch.int64)
            with torch.no_grad():
                seg, pred, _ = model(data, empty, empty, 0, 0)
            segs.append(seg.cpu().numpy()); preds.append(pred.cpu().numpy()); gts.append(np.stack([c[1] for c in batch]))
        gt = np.concatenate(gts, axis=0).reshape((-1,) + bbox.shape[1:])
        st.add_video(np.concatenate(segs, axis=0), gt, np.concatenate(preds, axis=0), label)
    return st
