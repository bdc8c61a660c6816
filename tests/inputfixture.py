"""Build-owned synthetic decoded videos + annotation records for the input-pipeline tests, in the
format /root/reference/datasets/ucf_dataloader.py consumes: frames [F,240,320,3] uint8 and
annotations [(start, end, label, [[x,y,w,h] per frame], [annotated frame ids], labeled_vid), ...]."""
import numpy as np


def case(k):
    """-> (frames or None, annotations, train flag).  Cases: 0 plain, 1 two annotations (extra random draw), 2 annotated frame near
    the start (skip falls back to 1), 3 annotated frame at the very start, 4 window clamped at the end, 5 single annotated frame,
    6 annotated frame beyond the video (zero sample), 7 no annotated frames (zero sample), 8 reader failure, 9 test split (centre crop)."""
    rng = np.random.default_rng(100 + k)
    F = [40, 36, 30, 28, 25, 33, 20, 22, 0, 31][k]
    frames = rng.integers(0, 256, (F, 240, 320, 3), dtype=np.uint8) if F else None

    def ann(s, e, label, frames_annot, lv):
        boxes = []
        x, y = int(rng.integers(0, 200)), int(rng.integers(0, 120))
        for f in range(s, e + 1):
            boxes.append([min(300, x + 3 * (f - s)), min(200, y + (f - s)), int(rng.integers(30, 120)), int(rng.integers(30, 100))])
        return (s, e, label, boxes, list(frames_annot), lv)
    if k == 0: a = [ann(5, 30, 7, [12, 20, 25], 1)]
    elif k == 1: a = [ann(2, 20, 3, [10, 15], 1), ann(18, 35, 3, [22, 30], 1)]
    elif k == 2: a = [ann(0, 20, 11, [5, 6], 0)]
    elif k == 3: a = [ann(0, 27, 2, [1, 2], 1)]
    elif k == 4: a = [ann(3, 24, 23, [22, 24], 1)]
    elif k == 5: a = [ann(4, 30, 0, [16], 0)]
    elif k == 6: a = [ann(0, 25, 5, [24], 1)]          # 24 >= 20 frames
    elif k == 7: a = [ann(0, 15, 9, [], 1)]
    elif k == 8: a = [ann(0, 10, 1, [4], 1)]
    else: a = [ann(6, 28, 14, [10, 20], 1)]
    return frames, a, k != 9


N_CASES = 10


def jhmdb_case(k):
    """-> (frames [F,256,256,3] uint8 or None, puppet masks [F,256,256,1] float64 with part ids 0..5, label, annot_frames, train).
    0: every frame annotated (the reference's 100 % setting), 1: sparse annotation (frames without truth inside the window),
    2: short video, window at the start with skip 1, 3: single annotated frame, 4: no annotated frame, 5: test split."""
    rng = np.random.default_rng(500 + k)
    F = [30, 34, 12, 26, 20, 28][k]
    frames = rng.integers(0, 256, (F, 256, 256, 3), dtype=np.uint8)
    masks = np.zeros((F, 256, 256, 1))
    for f in range(F):
        y, x = 60 + f, 40 + 2 * f
        masks[f, y:y + 90, x:x + 60, 0] = rng.integers(0, 6, (90, 60))
    ann = [np.arange(F), np.array([3, 9, 10, 17, 30]), np.array([1, 2, 5]), np.array([14]), np.array([], dtype=np.int64), np.arange(F)][k]
    return frames, masks, 3 + k, ann, k != 5


N_JHMDB = 6
