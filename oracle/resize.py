"""Oracle: cv2.resize on uint8 images (test infrastructure; numpy, CPU).

The reference's loaders call it three ways: `cv2.resize(frame, (256, 256), interpolation=cv2.INTER_AREA)` on the decoded
JHMDB frames (/root/reference/datasets/jhmdb_dataloader.py:252), `INTER_NEAREST` on the puppet masks (:267, :281) and
`INTER_LINEAR` on the 224x224 crop (:192, :208; ucf_dataloader.py:165, 171 -- the identity for the 224x224 frame size every
caller uses).  The arithmetic lives in a third-party dependency that is NOT in /root/reference and NOT installed in this image:
OpenCV (`opencv-python`, un-pinned in the reference's requirements.txt).  This file restates the published algorithm of
OpenCV 4.x `modules/imgproc/src/resize.cpp` for 8-bit images:

  * cv::resize: equal sizes -> copy; INTER_LINEAR at exactly 2x2 decimation is turned into INTER_AREA; "true" area
    interpolation only when BOTH axes shrink (scale >= 1), with the integer-scale fast path (`resizeAreaFast_`: 2x2 windows
    `(s0+s1+s2+s3+2) >> 2`, other windows round-half-even of `sum * (1.f / area)`) and the fractional-coverage path
    (`computeResizeAreaTab` + `ResizeArea_`: float32 accumulation in table order, `saturate_cast<uchar>` = round half even);
  * otherwise two taps per axis with 11-bit fixed-point coefficients (`INTER_RESIZE_COEF_BITS`): offsets / weights from the
    pixel-centre rule `(dx + 0.5) * scale - 0.5` for INTER_LINEAR, from the cell overlap `(dx+1) - (sx+1) * inv_scale` for
    INTER_AREA with an enlarging axis (the JHMDB case: 320 -> 256 shrinks, 240 -> 256 grows); horizontal pass in int32
    (`S[sx]*a0 + S[sx+1]*a1`, plain `S[sx]*2048` from the first column whose second tap would fall outside), vertical pass
    `((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2` with the row indices clipped when fetched;
  * INTER_NEAREST: `min(floor(dx * (1 / (dst/src))), src - 1)` per axis.

PARITY UNPINNED against OpenCV itself: there is no cv2 in this container to produce golden vectors, and the reference holds
no fixtures for this path.  What pins it instead: known answers that any correct restatement must satisfy
(tests/test_resize.py: identity, exact integer decimation means, constant images, monotone nearest indices, agreement within
one grey level with an independent float bilinear / area evaluation), and the HIP kernel is compared with this file bit for
bit."""
import numpy as np

INTER_NEAREST, INTER_LINEAR, INTER_AREA = 0, 1, 3
COEF_SCALE = 2048


def _sat_short(v):
    return np.clip(np.rint(np.asarray(v, np.float32)), -32768, 32767).astype(np.int32)


def _linear_axis(ssize, dsize, area_mode):
    """-> (ofs, coef[d,2] int32, float frac, first dst index whose second tap is outside) for one axis."""
    inv_scale = np.float64(dsize) / np.float64(ssize)
    scale = np.float64(1.0) / inv_scale
    d = np.arange(dsize, dtype=np.float64)
    if not area_mode:
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
    else:
        s = np.floor(d * scale).astype(np.int64)
        f = ((d + 1) - (s + 1) * inv_scale).astype(np.float32)
        f = np.where(f <= 0, np.float32(0), f - np.floor(f)).astype(np.float32)
    return s, f, scale


def resize_tables_linear(H, W, Ho, Wo, area_mode):
    sx, fx, _ = _linear_axis(W, Wo, area_mode)
    neg = sx < 0
    fx = np.where(neg, np.float32(0), fx); sx = np.where(neg, 0, sx)
    outside = sx + 1 >= W
    xmax = int(np.argmax(outside)) if outside.any() else Wo
    last = sx >= W - 1
    fx = np.where(last, np.float32(0), fx).astype(np.float32); sx = np.where(last, W - 1, sx)
    ia = np.stack([_sat_short((np.float32(1) - fx) * np.float32(COEF_SCALE)), _sat_short(fx * np.float32(COEF_SCALE))], 1)
    sy, fy, _ = _linear_axis(H, Ho, area_mode)
    ib = np.stack([_sat_short((np.float32(1) - fy) * np.float32(COEF_SCALE)), _sat_short(fy * np.float32(COEF_SCALE))], 1)
    return sx, ia, fx, xmax, sy, ib, fy


def area_tab(ssize, dsize):
    """computeResizeAreaTab -> list per dst index of (src index, float32 weight)."""
    scale = 1.0 / (np.float64(dsize) / np.float64(ssize))
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale; fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(np.ceil(fsx1)), int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1); sx1 = min(sx1, sx2)
        ent = []
        if sx1 - fsx1 > 1e-3:
            ent.append((sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            ent.append((sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            ent.append((sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
        tab.append(ent)
    return tab


def resize_kind(interpolation, H, W, Ho, Wo):
    sx, sy = 1.0 / (np.float64(Wo) / W), 1.0 / (np.float64(Ho) / H)
    ix, iy = int(np.rint(sx)), int(np.rint(sy))
    fast = abs(sx - ix) < np.finfo(np.float64).eps and abs(sy - iy) < np.finfo(np.float64).eps
    if (H, W) == (Ho, Wo):
        return "copy"
    if interpolation == INTER_NEAREST:
        return "nearest"
    if interpolation == INTER_LINEAR and fast and ix == 2 and iy == 2:
        interpolation = INTER_AREA
    if interpolation == INTER_AREA and sx >= 1 and sy >= 1:
        return "area_fast" if fast else "area"
    return "linear_area" if interpolation == INTER_AREA else "linear"


def resize(img, dsize, interpolation):
    """img uint8 [H, W] or [H, W, C]; dsize = (width, height) like cv2.resize -> uint8 [height, width(, C)]."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    squeeze = img.ndim == 2
    S = img[..., None] if squeeze else img
    H, W, C = S.shape
    Wo, Ho = int(dsize[0]), int(dsize[1])
    kind = resize_kind(interpolation, H, W, Ho, Wo)
    if kind == "copy":
        D = S.copy()
    elif kind == "nearest":
        ifx, ify = 1.0 / (np.float64(Wo) / W), 1.0 / (np.float64(Ho) / H)
        xs = np.minimum(np.floor(np.arange(Wo) * ifx).astype(np.int64), W - 1)
        ys = np.minimum(np.floor(np.arange(Ho) * ify).astype(np.int64), H - 1)
        D = S[ys][:, xs]
    elif kind == "area_fast":
        kx, ky = W // Wo, H // Ho
        win = S.reshape(Ho, ky, Wo, kx, C).astype(np.int64).sum(axis=(1, 3))
        if kx == 2 and ky == 2:
            D = ((win + 2) >> 2).astype(np.uint8)
        else:
            D = np.clip(np.rint(win.astype(np.float32) * np.float32(1.0 / (kx * ky))), 0, 255).astype(np.uint8)
    elif kind == "area":
        xt, yt = area_tab(W, Wo), area_tab(H, Ho)
        Sf = S.astype(np.float32)
        rows = {}          # horizontal pass per source row, float32, in table order

        def hrow(sy):
            if sy not in rows:
                buf = np.zeros((Wo, C), np.float32)
                for dx, ent in enumerate(xt):
                    acc = np.zeros(C, np.float32)
                    for si, al in ent:
                        acc = (acc + Sf[sy, si] * al).astype(np.float32)
                    buf[dx] = acc
                rows[sy] = buf
            return rows[sy]
        D = np.zeros((Ho, Wo, C), np.uint8)
        for dy, ent in enumerate(yt):
            acc = None
            for si, beta in ent:
                term = (beta * hrow(si)).astype(np.float32)
                acc = term if acc is None else (acc + term).astype(np.float32)
            D[dy] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    else:
        sx, ia, _fx, xmax, sy, ib, _fy = resize_tables_linear(H, W, Ho, Wo, kind == "linear_area")
        Si = S.astype(np.int64)
        r0 = np.clip(sy, 0, H - 1); r1 = np.clip(sy + 1, 0, H - 1)
        sx1 = np.minimum(sx + 1, W - 1)
        two = (np.arange(Wo) < xmax)[None, :, None]

        def hpass(R):
            a = R[:, sx] * ia[None, :, 0, None] + R[:, sx1] * ia[None, :, 1, None]
            return np.where(two, a, R[:, sx] * COEF_SCALE)
        h0, h1 = hpass(Si[r0]), hpass(Si[r1])
        b0, b1 = ib[:, 0][:, None, None].astype(np.int64), ib[:, 1][:, None, None].astype(np.int64)
        D = ((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2).astype(np.uint8)
    return D[..., 0] if squeeze else D


def resize_positive(mask, dsize):
    """`cv2.resize(m, dsize, interpolation=INTER_LINEAR) > 0` for a non-negative float mask (the reference's box masks are
    float64 {0, 1}, ucf_dataloader.py:170-172): positive wherever a tap with a positive weight meets a positive sample."""
    m = np.asarray(mask) > 0
    H, W = m.shape[:2]
    Wo, Ho = int(dsize[0]), int(dsize[1])
    if (H, W) == (Ho, Wo):
        return m.copy()
    if resize_kind(INTER_LINEAR, H, W, Ho, Wo) == "area_fast":          # exact 2x2 decimation runs as area: any sample of the window
        return m.reshape((Ho, 2, Wo, 2) + m.shape[2:]).any(axis=(1, 3))
    sx, _ia, fx, _xmax, sy, _ib, fy = resize_tables_linear(H, W, Ho, Wo, False)
    r0 = np.clip(sy, 0, H - 1); r1 = np.clip(sy + 1, 0, H - 1)
    sx1 = np.minimum(sx + 1, W - 1)
    px = (fx > 0)[None, :] if m.ndim == 2 else (fx > 0)[None, :, None]
    py = (fy > 0)[:, None] if m.ndim == 2 else (fy > 0)[:, None, None]
    h = lambda R: R[:, sx] | (px & R[:, sx1])
    return h(m[r0]) | (py & h(m[r1]))
