"""cv2.resize on uint8 images (the loaders' resizes): oracle/resize.py restates OpenCV's 8-bit algorithm
(cv2 is not in this image: parity against OpenCV itself is unpinned, see the oracle's header).  CPU tests pin the restatement
to known answers and the library's HOST table builder (pc_resize_tables) to it word for word; the GPU tests compare
pc_resize_u8 with the oracle bit for bit, incl. the JHMDB loader's 320x240 -> 256x256 case
(/root/reference/datasets/jhmdb_dataloader.py:252,267,281) and the loaders' INTER_LINEAR crop resize (:192,:208)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import resize as R
from picons_amd import capi

CASES = [  # (interpolation, H, W, Ho, Wo)
    (R.INTER_AREA, 240, 320, 256, 256),       # the JHMDB frames: x shrinks, y grows -> two-tap form with cell-overlap weights
    (R.INTER_NEAREST, 240, 320, 256, 256),    # the JHMDB puppet masks
    (R.INTER_LINEAR, 224, 224, 112, 112),     # exact 2x2 decimation: runs as area
    (R.INTER_LINEAR, 224, 224, 160, 160),     # crop -> smaller frame
    (R.INTER_LINEAR, 224, 224, 256, 300),     # crop -> larger frame
    (R.INTER_AREA, 240, 320, 96, 128),        # fractional coverage (2.5x)
    (R.INTER_AREA, 240, 320, 80, 80),         # integer scales 3 x 4
    (R.INTER_AREA, 250, 333, 100, 111),       # ragged fractional
    (R.INTER_AREA, 64, 64, 64, 64),           # copy
    (R.INTER_NEAREST, 37, 53, 90, 17),
]


def c_table(interp, H, W, Ho, Wo):
    L = capi.lib()
    n = L.pc_resize_tables(interp, H, W, Ho, Wo, None, 0)
    assert n >= 8
    a = np.zeros(n, np.int32)
    assert L.pc_resize_tables(interp, H, W, Ho, Wo, C.c_void_p(a.ctypes.data), n) == n
    return a


def fl(words):
    return np.asarray(words, np.int32).view(np.float32)


@pytest.mark.parametrize("interp,H,W,Ho,Wo", CASES)
def test_host_tables_match_the_oracle(interp, H, W, Ho, Wo):
    t = c_table(interp, H, W, Ho, Wo)
    kind = R.resize_kind(interp, H, W, Ho, Wo)
    assert ["nearest", "linear", "area", "area_fast", "copy"].index("linear" if kind == "linear_area" else kind) == t[0]
    if kind == "nearest":
        assert np.array_equal(t[t[1]:t[1] + Wo], np.minimum(np.floor(np.arange(Wo) * (1.0 / (Wo / W))).astype(int), W - 1))
        assert np.array_equal(t[t[3]:t[3] + Ho], np.minimum(np.floor(np.arange(Ho) * (1.0 / (Ho / H))).astype(int), H - 1))
    elif kind in ("linear", "linear_area"):
        sx, ia, fx, xmax, sy, ib, fy = R.resize_tables_linear(H, W, Ho, Wo, kind == "linear_area")
        assert np.array_equal(t[t[1]:t[1] + Wo], sx) and np.array_equal(t[t[2]:t[2] + 2 * Wo].reshape(Wo, 2), ia) and t[5] == xmax
        assert np.array_equal(t[t[3]:t[3] + Ho], sy) and np.array_equal(t[t[4]:t[4] + 2 * Ho].reshape(Ho, 2), ib)
        assert np.array_equal(fl(t[t[7]:t[7] + Wo]), fx) and np.array_equal(fl(t[t[7] + Wo:t[7] + Wo + Ho]), fy)
        assert np.all(ia.sum(1) == 2048) and np.all(ib.sum(1) == 2048)
    elif kind == "area":
        for off_s, off_a, ssz, dsz in ((t[1], t[2], W, Wo), (t[3], t[4], H, Ho)):
            tab = R.area_tab(ssz, dsz)
            start = t[off_s:off_s + dsz + 1]
            n = start[-1]
            si, al = t[off_s + dsz + 1:off_s + dsz + 1 + n], fl(t[off_a:off_a + n])
            assert n == sum(len(e) for e in tab)
            assert [int(v) for v in si] == [s for e in tab for s, _a in e]
            assert np.array_equal(al, np.array([a for e in tab for _s, a in e], np.float32))
            for d in range(dsz):                       # weights of a cell sum to 1
                assert abs(float(al[start[d]:start[d + 1]].sum()) - 1.0) < 1e-5
    elif kind == "area_fast":
        assert (t[5], t[6]) == (W // Wo, H // Ho)


def test_bad_arguments_are_refused():
    L = capi.lib()
    assert L.pc_resize_tables(2, 10, 10, 5, 5, None, 0) < 0           # INTER_CUBIC is not one of the loaders' modes
    assert L.pc_resize_tables(1, 0, 10, 5, 5, None, 0) < 0
    assert b"pc_resize_tables" in L.pc_last_error()


def _float_bilinear(img, Ho, Wo):
    """Independent float evaluation of the pixel-centre bilinear rule (torch, align_corners=False, no antialias)."""
    x = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
    y = torch.nn.functional.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False)
    return y[0].permute(1, 2, 0).numpy()


def test_oracle_known_answers():
    g = np.random.default_rng(5)
    img = g.integers(0, 256, (224, 224, 3), dtype=np.uint8)
    assert np.array_equal(R.resize(img, (224, 224), R.INTER_LINEAR), img)                         # the loaders' 224 -> 224: identity
    for interp in (R.INTER_NEAREST, R.INTER_LINEAR, R.INTER_AREA):                                # constant images stay constant
        for (H, W, Ho, Wo) in ((240, 320, 256, 256), (224, 224, 160, 160), (240, 320, 96, 128), (240, 320, 80, 80)):
            c = np.full((H, W, 3), 173, np.uint8)
            assert np.all(R.resize(c, (Wo, Ho), interp) == 173), (interp, H, W, Ho, Wo)
    # exact integer decimation = window mean with the documented rounding
    d2 = R.resize(img, (112, 112), R.INTER_AREA)
    win = img.reshape(112, 2, 112, 2, 3).astype(np.int64).sum(axis=(1, 3))
    assert np.array_equal(d2, ((win + 2) >> 2).astype(np.uint8)) and np.array_equal(d2, R.resize(img, (112, 112), R.INTER_LINEAR))
    img2 = g.integers(0, 256, (240, 320, 3), dtype=np.uint8)
    d34 = R.resize(img2, (80, 80), R.INTER_AREA)
    mean = img2.reshape(80, 3, 80, 4, 3).astype(np.float64).mean(axis=(1, 3))
    assert np.abs(d34.astype(np.float64) - mean).max() <= 0.5 + 1e-4
    # fractional coverage: within half a grey level of the exact area integral
    da = R.resize(img2, (128, 96), R.INTER_AREA).astype(np.float64)
    cx = np.zeros((128, 320)); cy = np.zeros((96, 240))
    for d in range(128):
        for s in range(320):
            cx[d, s] = max(0.0, min((d + 1) * 2.5, s + 1) - max(d * 2.5, s)) / 2.5
    for d in range(96):
        for s in range(240):
            cy[d, s] = max(0.0, min((d + 1) * 2.5, s + 1) - max(d * 2.5, s)) / 2.5
    exact = np.einsum("ds,swc,xw->dxc", cy, img2.astype(np.float64), cx)
    assert np.abs(da - exact).max() <= 0.5 + 2e-3
    # bilinear, both directions: within one grey level of an independent float evaluation
    for (Ho, Wo) in ((160, 160), (256, 300), (200, 131)):
        d = R.resize(img, (Wo, Ho), R.INTER_LINEAR).astype(np.float64)
        assert np.abs(d - _float_bilinear(img, Ho, Wo)).max() <= 1.0, (Ho, Wo)
    # the JHMDB frame resize: a smooth image stays within one grey level of the exact cell-overlap rule it emulates, in x
    # (area, 1.25x shrink) and of linear interpolation at the cell position in y (the 1.0667x enlargement)
    yy, xx = np.mgrid[0:240, 0:320]
    smooth = np.stack([(xx * 0.6 + yy * 0.2), (200 - xx * 0.3 + yy * 0.1), (yy * 0.9)], -1)
    sm8 = np.clip(np.rint(smooth), 0, 255).astype(np.uint8)
    dj = R.resize(sm8, (256, 256), R.INTER_AREA)
    assert dj.shape == (256, 256, 3)
    inner = dj[4:-4, 4:-4].astype(np.float64)
    ys = (np.arange(256) + 0.5) * (240 / 256) - 0.5; xs = (np.arange(256) + 0.5) * 1.25 - 0.5
    want = np.stack([(xs[None, :] * 0.6 + ys[:, None] * 0.2), (200 - xs[None, :] * 0.3 + ys[:, None] * 0.1), (ys[:, None] * 0.9 + 0 * xs[None, :])], -1)
    assert np.abs(inner - want[4:-4, 4:-4]).max() <= 1.5
    # nearest: indices are monotone, hit both ends, and masks stay {0,1}
    m = (g.random((240, 320)) < 0.3).astype(np.uint8)
    dn = R.resize(m, (256, 256), R.INTER_NEAREST)
    assert set(np.unique(dn)) <= {0, 1} and dn[0, 0] == m[0, 0] and dn[255, 255] == m[int(255 * 0.9375), int(255 * 1.25)]
    # positivity of a float-resized box mask
    box = np.zeros((224, 224)); box[50:90, 30:100] = 1.0
    pos = R.resize_positive(box, (160, 160))
    ref = _float_bilinear(box[..., None].astype(np.float32) * 255, 160, 160)[..., 0] > 0
    assert pos.shape == (160, 160) and np.array_equal(pos, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("interp,H,W,Ho,Wo", CASES)
def test_device_resize_matches_oracle(interp, H, W, Ho, Wo):
    from picons_amd import ops
    g = np.random.default_rng(H * 1000 + Wo)
    for Cc in (3, 1):
        img = g.integers(0, 256, (2, H, W, Cc), dtype=np.uint8)
        src = torch.from_numpy(img if Cc == 3 else img[..., 0].copy()).cuda()
        got = ops.resize_u8(src, Ho, Wo, interp).cpu().numpy()
        for i in range(2):
            want = R.resize(img[i] if Cc == 3 else img[i, :, :, 0], (Wo, Ho), interp)
            assert got[i].shape == want.shape and np.array_equal(got[i], want), (interp, H, W, Ho, Wo, Cc, i)
    if interp == R.INTER_LINEAR:
        m = (g.random((2, H, W)) < 0.2).astype(np.uint8)
        got = ops.resize_u8(torch.from_numpy(m).cuda(), Ho, Wo, interp, binarize=True).cpu().numpy()
        for i in range(2):
            assert np.array_equal(got[i].astype(bool), R.resize_positive(m[i].astype(np.float64), (Wo, Ho)))


@pytest.mark.gpu
def test_jhmdb_load_video_and_sample_on_device():
    """load_video's resizes (jhmdb_dataloader.py:252,267,281) + __getitem__ on the device against the oracle chain
    (oracle.resize then oracle.inputpipe), and a UCF sample at a frame size other than the crop (:165,:171)."""
    from oracle import inputpipe as oip
    from picons_amd import inputpipe as pip
    g = np.random.default_rng(11)
    F = 20
    frames = g.integers(0, 256, (F, 240, 320, 3), dtype=np.uint8)
    part = np.zeros((240, 320, F), np.uint8)
    for f in range(F):
        part[60 + f:150 + f, 100 + 2 * f:180 + 2 * f, f] = 1 + (f % 3)
    fr, mk, annot = pip.load_video_jhmdb(frames, part)
    want_f = np.stack([R.resize(frames[f], (256, 256), R.INTER_AREA) for f in range(F)])
    want_m = np.stack([R.resize(part[:, :, f], (256, 256), R.INTER_NEAREST) for f in range(F)])
    assert np.array_equal(fr.cpu().numpy(), want_f) and np.array_equal(mk.cpu().numpy(), want_m) and list(annot) == list(range(F))
    for train in (True, False):
        np.random.seed(3)
        got = pip.get_item_jhmdb(fr, mk, 7, annot, train=train)
        np.random.seed(3)
        want = oip.get_item_jhmdb(want_f, want_m, 7, annot, train=train)
        for k in ("data", "aug_data", "loc_msk", "mask_cls"):
            assert np.array_equal(got[k].cpu().numpy(), np.asarray(want[k], np.float32)), k
    # UCF sample at frame size 160: crop 224, INTER_LINEAR to 160, /255; mask by positivity
    from picons_amd import synthetic
    vid, ann = synthetic.make_decoded_video(4)
    np.random.seed(9)
    got = pip.get_item(vid, ann, train=True, size=160)
    np.random.seed(9)
    base = oip.get_item(vid, ann, train=True)           # the 224 sample with the same draws
    assert tuple(got["data"].shape) == (3, 8, 160, 160) and tuple(got["loc_msk"].shape) == (1, 8, 160, 160)
    d224 = np.asarray(base["data"])                      # [3,8,224,224] float64 = u8 / 255
    u8 = np.rint(d224 * 255).astype(np.uint8).transpose(1, 2, 3, 0)
    want = np.stack([R.resize(u8[t], (160, 160), R.INTER_LINEAR) for t in range(8)]).astype(np.float64) / 255.
    assert np.array_equal(got["data"].cpu().numpy(), want.transpose(3, 0, 1, 2).astype(np.float32))
    assert np.array_equal(got["aug_data"].cpu().numpy(), got["data"].cpu().numpy()[..., ::-1])
    wm = np.stack([R.resize_positive(np.asarray(base["loc_msk"])[0, t], (160, 160)) for t in range(8)])
    assert np.array_equal(got["loc_msk"].cpu().numpy()[0].astype(bool), wm)
