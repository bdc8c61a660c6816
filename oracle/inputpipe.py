"""CPU oracle (TEST INFRASTRUCTURE ONLY): the reference's per-sample input preparation restated in numpy from
/root/reference/datasets/ucf_dataloader.py -- the box rasterisation of `load_video` (:204-221, :260-264) and
`__getitem__` (:84-191).  Decoding (skvideo `vread`, :194) is outside: the functions take the decoded uint8 frames.

Pinned: tests/golden/input_pipe.npz holds outputs of the reference's own `UCF101DataLoader.__getitem__` (video reader
stubbed with the synthetic clips of tests/inputfixture.py; `cv2.resize` stubbed as the identity it is when a 224x224
crop is resized to 224x224, :156,:162) -- tools/make_input_golden.py; tests/test_inputpipe.py checks this file against it.
Random draws go through the global `np.random` in the reference's order, so a seed reproduces its choices."""
import numpy as np
import torch

DEPTH = 8


def rasterise(annotations, n_frames, h, w):
    """load_video :204-221 (+ the unused draw :213-214): bbox [F,H,W,1] uint8, label, annotated frames, labeled flag.
    annotations: [(start_frame, end_frame, label, boxes[x,y,w,h] per frame, annotated frame ids, labeled_vid), ...]."""
    bbox = np.zeros((n_frames, h, w, 1), dtype=np.uint8)
    label, labeled_vid = -1, -1
    if len(annotations) > 1:
        np.random.randint(0, len(annotations))                       # :213-214 (result unused by the caller's outputs)
    multi = []
    for ann in annotations:
        multi.extend(ann[4])
        start_frame, end_frame, label, labeled_vid = ann[0], ann[1], ann[2], ann[5]
        for f in range(start_frame, min(n_frames, end_frame + 1)):
            x, y, bw, bh = ann[3][f - start_frame]
            bbox[f, y:y + bh, x:x + bw, :] = 1
    return bbox, label, list(set(multi)), labeled_vid


def _empty(height, width):
    z = np.zeros((DEPTH, height, width, 3)); m = np.zeros((DEPTH, height, width, 1))
    v = torch.from_numpy(np.transpose(z, [3, 0, 1, 2])); lm = torch.from_numpy(np.transpose(m, [3, 0, 1, 2]))
    return {'data': v, 'loc_msk': lm, 'action': torch.Tensor([0]), 'aug_data': v, 'label_vid': 0}


def get_item(clip, annotations, train=True, height=224, width=224):
    """__getitem__ :84-191 on decoded frames `clip` [F,H,W,3] uint8 (None: the reader failed)."""
    if clip is None:
        return _empty(height, width)
    bbox_clip, label, annot_frames, labeled_vid = rasterise(annotations, clip.shape[0], clip.shape[1], clip.shape[2])
    vlen, clip_h, clip_w, _ = clip.shape
    vskip = 2
    if len(annot_frames) == 1:
        sel = annot_frames[0]
    else:
        if len(annot_frames) <= 0:
            return _empty(height, width)
        sel = annot_frames[np.random.randint(0, len(annot_frames))]
    start = sel - int((DEPTH * vskip) / 2)
    if start < 0:
        vskip = 1
        start = sel - int((DEPTH * vskip) / 2)
        if start < 0:
            start = 0
            vskip = 1
    if sel >= vlen:
        return _empty(height, width)
    if start + (DEPTH * vskip) >= vlen:
        start = vlen - (DEPTH * vskip)
    span = np.arange(DEPTH) * vskip + start
    video = clip[span]; boxes = bbox_clip[span]
    if train:
        h0 = np.random.randint(0, clip_h - 224); w0 = np.random.randint(0, clip_w - 224)
    else:
        h0 = int((clip_h - 224) / 2); w0 = int((clip_w - 224) / 2)
    video_rgb = np.zeros((DEPTH, height, width, 3)); label_cls = np.zeros((DEPTH, height, width, 1))
    for j in range(DEPTH):
        video_rgb[j] = video[j][h0:h0 + 224, w0:w0 + 224, :] / 255.      # cv2.resize 224 -> 224 is the identity (:156)
        bb = boxes[j][h0:h0 + 224, w0:w0 + 224, 0]
        label_cls[j, bb > 0, 0] = 1.
    flip = video_rgb[:, :, ::-1, :]
    return {'data': torch.from_numpy(np.transpose(video_rgb, [3, 0, 1, 2])), 'loc_msk': torch.from_numpy(np.transpose(label_cls, [3, 0, 1, 2])),
            'action': torch.Tensor([label]), 'aug_data': torch.from_numpy(np.transpose(flip, [3, 0, 1, 2]).copy()), 'label_vid': labeled_vid}


def get_item_jhmdb(clip, bbox_clip, label, annot_frames, train=True, height=224, width=224):
    """/root/reference/datasets/jhmdb_dataloader.py:102-230 after `load_video` (:232-310: cv2 decode + resize + .mat read, not
    restated): clip [F,H,W,3] frames (uint8 values), bbox_clip [F,H,W] or [F,H,W,1] puppet masks (> 0 foreground), the class id
    and the frames that carry truth.  Only frames in `annot_frames` (or, at skip 2, whose successor is) get a mask; `mask_cls`
    marks them."""
    def empty():
        z = torch.from_numpy(np.transpose(np.zeros((DEPTH, height, width, 3)), [3, 0, 1, 2]))
        m = torch.from_numpy(np.transpose(np.zeros((DEPTH, height, width, 1)), [3, 0, 1, 2]))
        return {'data': z, 'loc_msk': m, 'action': torch.Tensor([0]), 'mask_cls': m.clone(), 'aug_data': z}
    bbox_clip = np.reshape(bbox_clip, (bbox_clip.shape[0], bbox_clip.shape[1], bbox_clip.shape[2], 1))      # :113 (before the None test, as there)
    if clip is None:
        return empty()
    vlen, clip_h, clip_w, _ = clip.shape
    vskip = 2
    if len(annot_frames) == 1:
        sel = annot_frames[0]
    else:
        if len(annot_frames) <= 0:
            return empty()
        sel = annot_frames[np.random.randint(0, len(annot_frames))]
    start = sel - int((DEPTH * vskip) / 2)
    if start < 0:
        vskip = 1
        start = sel - int((DEPTH * vskip) / 2)
        if start < 0:
            start, vskip = 0, 1
    if sel >= vlen:
        return empty()
    if start + (DEPTH * vskip) >= vlen:
        start = vlen - (DEPTH * vskip)
    span = np.arange(DEPTH) * vskip + start
    video = clip[span]; boxes = bbox_clip[span]
    if train:
        h0 = np.random.randint(0, clip_h - 224); w0 = np.random.randint(0, clip_w - 224)
    else:
        h0 = int((clip_h - 224) / 2); w0 = int((clip_w - 224) / 2)
    video_rgb = np.zeros((DEPTH, height, width, 3)); label_cls = np.zeros((DEPTH, height, width, 1)); mask_cls = np.zeros((DEPTH, height, width, 1))
    for j in range(DEPTH):
        video_rgb[j] = video[j][h0:h0 + 224, w0:w0 + 224, :] / 255.
        valid = (span[j] in annot_frames or span[j] + 1 in annot_frames) if vskip == 2 else (span[j] in annot_frames)     # :187-194
        if valid:
            bb = boxes[j][h0:h0 + 224, w0:w0 + 224, 0]
            label_cls[j, bb > 0, 0] = 1.
            mask_cls[j] = 1.
    flip = video_rgb[:, :, ::-1, :]
    tr = lambda a: torch.from_numpy(np.transpose(a, [3, 0, 1, 2]))
    return {'data': tr(video_rgb), 'loc_msk': tr(label_cls), 'action': torch.Tensor([label]), 'mask_cls': tr(mask_cls),
            'aug_data': torch.from_numpy(np.transpose(flip, [3, 0, 1, 2]).copy())}
