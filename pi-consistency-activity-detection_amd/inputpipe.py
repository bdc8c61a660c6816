"""Per-sample input preparation with the pixel work on the device.

Mirrors /root/reference/datasets/ucf_dataloader.py after decoding: `load_video`'s box bookkeeping (:204-264) and
`__getitem__` (:84-191) -- which annotated frame, which 8 frames (skip 2, falling back to 1 near the start, clamped at the
end), which 224x224 crop (random for training, centre otherwise), frame / 255, the horizontally flipped copy, the
foreground mask from the per-frame boxes.  The reference does all of it in numpy on float64 host arrays (9.6 MB per clip
and copy); here the host only makes the integer decisions, in the reference's order of `np.random` draws (so a seed
reproduces its choices), uploads the 8 selected uint8 frames (1.8 MB) and `pc_clip_from_u8` writes the fp32 NCDHW tensors
the step consumes (csrc/inputpipe.hip).  `cv2.resize` of a 224x224 crop to 224x224 (:156,:162) is the identity and is not
performed; for another frame size (`size=`) the crop is resized on the device with cv2's 8-bit INTER_LINEAR arithmetic and the
box mask with the positivity rule of its float path (pc_resize_u8).  The JHMDB loader's own resizes -- every decoded frame
to 256x256 with INTER_AREA, every puppet mask with INTER_NEAREST (jhmdb_dataloader.py:252,267,281) -- are `load_video_jhmdb`.
No CPU path for the pixel work: the ops raise without the HIP library."""
import numpy as np
import torch

from . import ops

DEPTH = 8
CROP = 224


class _PinnedRing:
    """Page-locked staging buffers for the 8 selected frames of a sample (1.8 MB at 240x320): the gather out of the decoded video is the
    host's one copy, the upload is an asynchronous DMA on the caller's stream.  (Through pageable memory the same upload blocked the host
    for ~2.5 ms per sample -- the driver stages it in chunks -- which made the per-step input work longer than the step itself.)  A buffer
    is reused only after the copy that last read it has finished (event per slot)."""

    def __init__(self, depth=24):
        self.depth, self.buf, self.ev, self.nxt = depth, {}, {}, {}

    def _slot(self, shape, dtype):
        """Round robin over `depth` buffers PER (shape, dtype): three steps of slack at eight samples a step.  Buffers are allocated on first
        use only (page-locking memory synchronises the device)."""
        kind = (tuple(shape), str(dtype))
        i = self.nxt.get(kind, 0)
        self.nxt[kind] = (i + 1) % self.depth
        key = kind + (i,)
        if key not in self.buf:
            self.buf[key] = torch.empty(tuple(shape), dtype=torch.from_numpy(np.empty(0, dtype)).dtype).pin_memory()
            self.ev[key] = None
        if self.ev[key] is not None:
            self.ev[key].synchronize()
        return key, self.buf[key]

    def _send(self, key, b, device):
        d = b.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(d.device))
        self.ev[key] = ev
        return d

    def upload(self, frames, span, device):
        key, b = self._slot((len(span),) + tuple(frames.shape[1:]), frames.dtype)
        bn = b.numpy()
        for t, f in enumerate(span):          # eight contiguous frame copies (np.take's generic gather is 10x slower)
            bn[t] = frames[int(f)]
        return self._send(key, b, device)

    def upload_array(self, arr, device):
        """A small host array (the box rectangles): through pageable memory even 1 KB is a BLOCKING copy that waits for everything queued
        on the stream -- with the sample kernels of a busy side stream in front of it, 1-2.5 ms per sample."""
        key, b = self._slot(arr.shape, arr.dtype)
        b.numpy()[...] = arr
        return self._send(key, b, device)


_ring = _PinnedRing()


def pinned_upload(arr, device):
    """Host numpy array -> device tensor through the page-locked ring, asynchronously on the current stream (StepEngine.stage's per-step
    scalars: class ids, labeled flags, shuffle index, Dropout3d draws)."""
    return _ring.upload_array(np.ascontiguousarray(arr), device)


def frame_boxes(annotations, n_frames):
    """load_video :204-221: per frame the boxes drawn into `bbox`, plus label, the annotated frame ids and the labeled flag.
    Consumes the draw of :213-214 like the reference."""
    if len(annotations) > 1:
        np.random.randint(0, len(annotations))
    per_frame = {}
    multi, label, labeled_vid = [], -1, -1
    for ann in annotations:
        multi.extend(ann[4])
        start_frame, end_frame, label, labeled_vid = ann[0], ann[1], ann[2], ann[5]
        for f in range(start_frame, min(n_frames, end_frame + 1)):
            per_frame.setdefault(f, []).append(ann[3][f - start_frame])
    return per_frame, label, list(set(multi)), labeled_vid


def frame_meta(annotations):
    """frame_boxes without the per-frame table (a Python loop over every annotated frame of the video: 1 ms per sample): label, annotated
    frame ids, labeled flag; consumes the same draw."""
    if len(annotations) > 1:
        np.random.randint(0, len(annotations))
    multi, label, labeled_vid = [], -1, -1
    for ann in annotations:
        multi.extend(ann[4])
        label, labeled_vid = ann[2], ann[5]
    return label, list(set(multi)), labeled_vid


def boxes_of(annotations, n_frames, frame_ids):
    """The rows of frame_boxes' per-frame table for `frame_ids` only (same order of boxes: annotation order)."""
    out = {}
    for f in frame_ids:
        f = int(f)
        for ann in annotations:
            if ann[0] <= f <= ann[1] and f < n_frames:
                out.setdefault(f, []).append(ann[3][f - ann[0]])
    return out


def choose_window(annot_frames, vlen):
    """__getitem__ :107-143 -> 8 frame ids, or None for the all-zero sample."""
    vskip = 2
    if len(annot_frames) == 1:
        sel = annot_frames[0]
    else:
        if len(annot_frames) <= 0:
            return None
        sel = annot_frames[np.random.randint(0, len(annot_frames))]
    start = sel - int((DEPTH * vskip) / 2)
    if start < 0:
        vskip = 1
        start = sel - int((DEPTH * vskip) / 2)
        if start < 0:
            start, vskip = 0, 1
    if sel >= vlen:
        return None
    if start + (DEPTH * vskip) >= vlen:
        start = vlen - (DEPTH * vskip)
    return np.arange(DEPTH) * vskip + start


def _empty(device, size):
    z = torch.zeros(3, DEPTH, size, size, device=device)
    return {'data': z, 'loc_msk': torch.zeros(1, DEPTH, size, size, device=device), 'action': torch.Tensor([0]), 'aug_data': z, 'label_vid': 0}


def _resized_sample(video8, mask8, size):
    """The per-frame `cv2.resize(img, (size, size), INTER_LINEAR)` of __getitem__ (:165,:171) for a frame size other than the
    crop: video8 uint8 [8,224,224,3], mask8 uint8 {0,1} [8,224,224] on the device -> data, aug [3,8,size,size], mask [8,size,size]."""
    v = ops.resize_u8(video8, size, size, 1)
    m = ops.resize_u8(mask8, size, size, 1, binarize=True)
    data = (v.to(torch.float64) / 255.).to(torch.float32).permute(3, 0, 1, 2).contiguous()      # img / 255. in float64, then fp32
    return data, torch.flip(data, [3]), m.to(torch.float32)


def get_item(frames, annotations, train=True, device="cuda", size=CROP, out=None, ndhwc4=False):
    """One sample as `UCF101DataLoader.__getitem__` returns it, with fp32 device tensors instead of float64 host tensors
    (the train loop casts to `torch.cuda.FloatTensor` first thing, main_ucf101.py:52-56).  frames: decoded uint8 [F,H,W,3]
    (numpy, or a device tensor when the decoder already writes to HBM), None when the reader failed (:90-98).
    size: the loader's frame size `[h, w]` (square); every caller of the reference uses 224, where the resize is the identity.
    out: (data [3,8,S,S], aug_data [3,8,S,S], loc_msk [1,8,S,S]) float32 device views to write the sample into -- its place in a minibatch
    staging buffer (StepEngine.sample_stager) -- instead of fresh tensors; the returned dict then holds those views.
    ndhwc4 (with out, size 224): data / aug_data views are [8,S,S,4] and are written as (r, g, b, 0) per position -- the layout the network's
    first conv reads -- instead of the loader's [3,8,S,S]."""
    if ndhwc4 and (out is None or size != CROP):
        raise ValueError("get_item: ndhwc4 is the staging layout of StepEngine.sample_stager (out=..., size 224)")

    def _into_out(e):
        if out is not None:
            for k, t in zip(("data", "aug_data", "loc_msk"), out):
                src = e[k]
                if ndhwc4 and k != "loc_msk":
                    t.zero_(); t[..., :3].copy_(src.permute(1, 2, 3, 0))
                else:
                    t.copy_(src.reshape(t.shape))
                e[k] = t
        return e
    if frames is None:
        return _into_out(_empty(device, size))
    vlen, clip_h, clip_w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    label, annot_frames, labeled_vid = frame_meta(annotations)
    span = choose_window(annot_frames, vlen)
    if span is None:
        return _into_out(_empty(device, size))
    per_frame = boxes_of(annotations, vlen, span)
    if train:
        h0 = np.random.randint(0, clip_h - CROP); w0 = np.random.randint(0, clip_w - CROP)       # :146-149
    else:
        h0 = int((clip_h - CROP) / 2); w0 = int((clip_w - CROP) / 2)
    # boxes of the selected frames, clipped the way numpy clips bbox[f, y:y+h, x:x+w] (:218)
    R = max([len(per_frame.get(int(f), [])) for f in span] + [1])
    rects = np.zeros((DEPTH, R, 4), np.int32)
    for t, f in enumerate(span):
        for r, (x, y, bw, bh) in enumerate(per_frame.get(int(f), [])):
            x0, x1, _ = slice(x, x + bw).indices(clip_w); y0, y1, _ = slice(y, y + bh).indices(clip_h)
            rects[t, r] = (x0, x1, y0, y1)
    if torch.is_tensor(frames):
        video, ids = frames.to(device), span
    else:
        video, ids = _ring.upload(frames, span, device), np.arange(DEPTH)   # only the 8 frames travel (pinned staging, asynchronous)
    rects_d = _ring.upload_array(rects, device)
    if size != CROP:
        crop8 = video.contiguous()[torch.as_tensor(np.asarray(ids), device=video.device).long(), h0:h0 + CROP, w0:w0 + CROP].contiguous()
        _d, _a, m224 = ops.clip_from_u8(video.contiguous(), ids, h0, w0, rects_d, CROP)          # the box mask of the crop
        data, aug, mask = _resized_sample(crop8, m224.to(torch.uint8), size)
        if out is not None:
            for src, t in zip((data, aug, mask), out):
                t.copy_(src.reshape(t.shape))
            data, aug, mask = out
    else:
        data, aug, mask = ops.clip_from_u8(video.contiguous(), ids, h0, w0, rects_d, CROP,
                                           out=None if out is None else (out[0], out[1], out[2].view(DEPTH, size, size)), ndhwc4=ndhwc4)
    return {'data': data, 'loc_msk': mask.view(1, DEPTH, size, size), 'action': torch.Tensor([label]), 'aug_data': aug, 'label_vid': labeled_vid}


def load_video_jhmdb(frames, part_mask, device="cuda"):
    """The pixel work of the JHMDB loader's `load_video` (jhmdb_dataloader.py:236-283) on the device: every decoded frame
    (uint8 [F,240,320,3] as cv2.VideoCapture yields them) resized to 256x256 with cv2.INTER_AREA (:252), every puppet mask
    (`part_mask` uint8 [240,320,M] from the .mat file) with cv2.INTER_NEAREST (:267,:281).  -> (frames uint8 [F,256,256,3],
    masks uint8 [M,256,256], annot_frames = arange(M)) -- what get_item_jhmdb consumes (device tensors stay on the device)."""
    f = (frames if torch.is_tensor(frames) else torch.from_numpy(np.ascontiguousarray(frames))).to(device).contiguous()
    pm = np.ascontiguousarray(np.transpose(np.asarray(part_mask), (2, 0, 1))) if not torch.is_tensor(part_mask) else part_mask.permute(2, 0, 1).contiguous()
    m = (pm if torch.is_tensor(pm) else torch.from_numpy(pm)).to(device).to(torch.uint8).contiguous()
    return ops.resize_u8(f, 256, 256, 3), ops.resize_u8(m, 256, 256, 0), np.arange(m.shape[0])


def get_item_jhmdb(frames, masks, label, annot_frames, train=True, device="cuda"):
    """One sample as /root/reference/datasets/jhmdb_dataloader.py `__getitem__` (:102-230) returns it after `load_video`:
    frames uint8 [F,H,W,3] (the loader's 256x256 resized frames), masks [F,H,W] or [F,H,W,1] puppet masks (> 0 foreground;
    any dtype, compared against 0 on the host when they are not uint8), the class id and the frames that carry truth.  Keys:
    data, loc_msk, action, mask_cls, aug_data."""
    def empty():
        z = torch.zeros(3, DEPTH, CROP, CROP, device=device); m = torch.zeros(1, DEPTH, CROP, CROP, device=device)
        return {'data': z, 'loc_msk': m, 'action': torch.Tensor([0]), 'mask_cls': m.clone(), 'aug_data': z}
    if frames is None:
        return empty()
    vlen, clip_h, clip_w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    span = choose_window(list(annot_frames) if not isinstance(annot_frames, list) else annot_frames, vlen)
    if span is None:
        return empty()
    vskip = int(span[1] - span[0])
    if train:
        h0 = np.random.randint(0, clip_h - CROP); w0 = np.random.randint(0, clip_w - CROP)
    else:
        h0 = int((clip_h - CROP) / 2); w0 = int((clip_w - CROP) / 2)
    af = set(int(a) for a in annot_frames)
    valid = [(int(f) in af or int(f) + 1 in af) if vskip == 2 else (int(f) in af) for f in span]               # :187-194
    idx = torch.as_tensor(np.asarray(span)).long()
    if torch.is_tensor(masks):          # load_video_jhmdb's device tensors: only the 8 frames are gathered, nothing travels
        m8d = masks.reshape(vlen, clip_h, clip_w)[idx.to(masks.device)].to(device)
        m8d = (m8d if m8d.dtype == torch.uint8 else (m8d > 0).to(torch.uint8)).contiguous()
    else:
        m = np.asarray(masks).reshape(vlen, clip_h, clip_w)[span]
        m8d = torch.from_numpy(np.ascontiguousarray(m if m.dtype == np.uint8 else (m > 0).astype(np.uint8))).to(device)
    if torch.is_tensor(frames):
        video = frames[idx.to(frames.device)].to(device).contiguous()
    else:
        video = torch.from_numpy(np.ascontiguousarray(np.asarray(frames)[span])).to(device)
    data, aug, mask, mask_cls = ops.clip_from_u8_masks(video, np.arange(DEPTH), h0, w0, m8d, valid, CROP)
    return {'data': data, 'loc_msk': mask.view(1, DEPTH, CROP, CROP), 'action': torch.Tensor([label]), 'mask_cls': mask_cls.view(1, DEPTH, CROP, CROP),
            'aug_data': aug}
