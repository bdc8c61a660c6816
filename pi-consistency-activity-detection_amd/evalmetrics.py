"""f-mAP / v-mAP evaluation with the accumulation on the device.

Mirrors /root/reference/evaluate_ucf101.py:73-186 (evaluate_jhmdb.py: same loop, 21 classes): every video is cut
into 8-frame clips (both phases of the frame skip, zero frames past the end, clips without truth dropped), batches
of 14 clips go through the network in eval mode, masks are sigmoid >= 0.5, and per class the frames / videos whose
intersection-over-union reaches k/20 are counted for k = 0..19.  The reference pulls every mask to the host and
loops over frames in numpy; here the masks never leave HBM: `pc_seg_frame_counts` reduces them to three integers per
frame and `pc_map_accumulate` adds a video into int32 tables (csrc/evalmetrics.hip).  No CPU path: the ops raise
without the HIP library."""
import numpy as np
import torch

from . import ops

N_THR = 20


def make_clips(video, bbox, f_skip=2):
    """evaluate_ucf101.py:79-97.  video [F,H,W,3], bbox [F,H,W,1] (numpy) -> clips [n,8,H,W,3], boxes [n,8,H,W,1] (float32):
    frame k of clip (i, j) is frame i + j + k*f_skip, i = 0, 16, 32.., j = 0..f_skip-1, zero past the end; clips whose
    boxes are all zero are dropped."""
    video = np.asarray(video, np.float32); bbox = np.asarray(bbox, np.float32)
    F = video.shape[0]
    starts = (np.arange(0, F, 8 * f_skip)[:, None] + np.arange(f_skip)[None, :]).reshape(-1)
    ind = starts[:, None] + np.arange(8)[None, :] * f_skip                     # [n, 8]
    ok = ind < F
    safe = np.where(ok, ind, 0)
    v = video[safe] * ok[:, :, None, None, None]
    b = bbox[safe] * ok[:, :, None, None, None]
    keep = b.reshape(b.shape[0], -1).sum(1) != 0
    return v[keep], b[keep]


def make_clips_device(video, bbox, f_skip=2, device="cuda"):
    """make_clips with the frame gather, the zero frames and the NCDHW permutation on the device: the video goes up once as it
    is (each frame belongs to exactly one clip, so nothing is uploaded twice) and the host does no pass over the pixels.
    -> data [n,3,8,H,W], boxes [n,8,H,W] (device float32), same clips in the same order as make_clips."""
    v = torch.as_tensor(video, dtype=torch.float32).to(device, non_blocking=True)
    b = torch.as_tensor(bbox, dtype=torch.float32).to(device, non_blocking=True)
    F = v.shape[0]
    starts = (np.arange(0, F, 8 * f_skip)[:, None] + np.arange(f_skip)[None, :]).reshape(-1)
    ind = starts[:, None] + np.arange(8)[None, :] * f_skip
    ok = torch.from_numpy(ind < F).to(device)
    safe = torch.from_numpy(np.where(ind < F, ind, 0)).to(device)
    bx = b.reshape(F, b.shape[1], b.shape[2])[safe] * ok[:, :, None, None]                 # [n,8,H,W]
    keep = bx.flatten(1).sum(1) != 0                                                         # clips without truth are dropped (:95-96)
    safe, ok, bx = safe[keep], ok[keep], bx[keep]
    data = (v[safe] * ok[:, :, None, None, None]).permute(0, 4, 1, 2, 3).contiguous()        # [n,3,8,H,W]
    return data, bx.contiguous()


class MapAccumulator:
    """The accumulators of evaluate_ucf101.py:66-72 as int32 device tables."""

    def __init__(self, n_classes=24, device="cuda"):
        self.n_classes = n_classes
        z = lambda *s: torch.zeros(*s, dtype=torch.int32, device=device)
        self.frame_hits, self.video_hits = z(n_classes, N_THR), z(n_classes, N_THR)
        self.n_frames, self.n_vids, self.n_correct = z(n_classes), z(n_classes), z(1)

    def add_video(self, seg_logits, gt, predictions, label):
        """seg_logits (B,1,8,H,W) device fp32 logits of one video's clips, gt (B,8,H,W[,1]) truth frames in the same order,
        predictions (B, n_classes) (evaluate_ucf101.py:128-183)."""
        counts = ops.seg_frame_counts(seg_logits.contiguous(), gt.reshape(seg_logits.shape).contiguous())
        ops.map_accumulate(counts, label, self.frame_hits, self.video_hits, self.n_frames, self.n_vids)
        self.n_correct += (predictions.float().mean(0).argmax() == int(label)).to(torch.int32)     # :140-144

    def result(self):
        """evaluate_ucf101.py:181-186: float64 ratios on the host; a class without a video gives NaN there and here."""
        fh, vh = self.frame_hits.cpu().numpy().astype(np.float64), self.video_hits.cpu().numpy().astype(np.float64)
        nf, nv = self.n_frames.cpu().numpy().astype(np.float64)[:, None], self.n_vids.cpu().numpy().astype(np.float64)[:, None]
        with np.errstate(invalid="ignore", divide="ignore"):
            return dict(accuracy=float(self.n_correct.item()) / nv.sum(), fmAP=np.mean(fh / nf, axis=0), vmAP=np.mean(vh / nv, axis=0),
                        frame_ious=fh, video_ious=vh, n_tot_frames=nf, n_vids=nv, n_correct=int(self.n_correct.item()))


def evaluate(model, videos, n_classes=24, clip_batch_size=14, device="cuda", pack=False):
    """One checkpoint over an iterable of (video, bbox, label) (evaluate_ucf101.py:73-186).  `model` is called as the
    reference calls it: model(data (B,3,8,H,W), empty_action, empty_action, 0, 0) -> (logits, class scores, _).
    pack=False: batches of up to `clip_batch_size` clips of ONE video, as the reference forms them.  pack=True: clips of
    consecutive videos share batches (always full ones); in eval mode every clip's outputs are independent of what else is
    in the batch, so the tables are the same -- only the under-filled 2-4-clip launches of short videos disappear."""
    acc = MapAccumulator(n_classes, device)
    empty_for = lambda n: torch.full((n, 1), 500, dtype=torch.int64, device=device)                  # :121-122
    with torch.no_grad():
        if not pack:
            for video, bbox, label in videos:
                clips, boxes = make_clips_device(video, bbox, device=device)
                if clips.shape[0] == 0:
                    continue                                           # "Video has no bounding boxes" (:99-101)
                segs, preds = [], []
                for i in range(0, clips.shape[0], clip_batch_size):
                    data = clips[i:i + clip_batch_size]
                    seg, pred, _ = model(data, empty_for(data.shape[0]), empty_for(data.shape[0]), 0, 0)
                    segs.append(seg); preds.append(pred)
                acc.add_video(torch.cat(segs, 0), boxes, torch.cat(preds, 0), int(label))
            return acc
        queue, owners = [], []                  # clips waiting for a batch, (video record, index inside the video) per clip
        open_videos = []                        # records in arrival order: [label, boxes, n clips, segs, preds]

        def run(batch, who):
            data = torch.stack(batch)
            seg, pred, _ = model(data, empty_for(len(batch)), empty_for(len(batch)), 0, 0)
            for k, (rec, _i) in enumerate(who):
                rec[3].append(seg[k]); rec[4].append(pred[k])
            while open_videos and len(open_videos[0][3]) == open_videos[0][2]:        # all clips of the oldest video are back
                lab, boxes, _n, segs, preds = open_videos.pop(0)
                acc.add_video(torch.stack(segs), boxes, torch.stack(preds), lab)
        for video, bbox, label in videos:
            clips, boxes = make_clips_device(video, bbox, device=device)
            if clips.shape[0] == 0:
                continue
            rec = [int(label), boxes, int(clips.shape[0]), [], []]
            open_videos.append(rec)
            for i in range(clips.shape[0]):
                queue.append(clips[i]); owners.append((rec, i))
                if len(queue) == clip_batch_size:
                    run(queue, owners); queue, owners = [], []
        if queue:
            run(queue, owners)
    return acc
