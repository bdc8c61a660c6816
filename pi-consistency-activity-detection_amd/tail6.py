"""Host side of the merged decoder tail (csrc/tail6.hip, capsules_ucf101.py:504-509): the per-class descriptors.

Input positions fall into 8 classes z = 4*(it == 0) + 2*(ih == 0) + (iw == 0); every class is a rectangular
sub-lattice of the [It][Ih][Iw] grid (index 0, or indices 1..I-1, per dimension) and has its own weight matrix."""
from . import desc as D

SP = 128          # 125 column slots (k5t, k5h, k5w) padded to one 128-wide tile
NSLOT = 125       # the slots that carry work (FLOP booking)


def classes(thw):
    """-> [(z, start[3], extent[3])] of the non-empty classes."""
    out = []
    for z in range(8):
        first = [(z >> 2) & 1, (z >> 1) & 1, z & 1]
        start = [0 if f else 1 for f in first]
        ext = [1 if f else thw[d] - 1 for d, f in enumerate(first)]
        if min(ext) >= 1:
            out.append((z, start, ext))
    return out


def _sub(d, start, ext):
    e = dict(d)
    e.update(Tq=ext[0], Hq=ext[1], Wq=ext[2], ooff=list(start), ioff0=list(start))
    return e


def conv_descs(N, thw, Ci, ldx):
    """colsT[n][:][i] = x[n][i][:] . W5f[n][z], written channel-major ([n][slot][position], PC_F_TOUT) for the streaming gather:
    [(z, desc)], weights at W5f + z*SP*Ci, per-sample stride 8*SP*Ci."""
    from . import capi
    base = D.conv_fwd(N, thw, Ci, ldx, SP, SP, (1, 1, 1), (1, 1, 1), (0, 0, 0), thw, flags=capi.F_TOUT, groups=N)
    base["wgstride"] = 8 * SP * Ci
    base["Co_real"] = NSLOT
    return [(z, _sub(base, s, e)) for z, s, e in classes(thw)]


def dgrad_descs(N, thw, Ci, lddx, accum):
    """dx[n][i][:] (+)= dcols[n][i][:] . W5t[n][z]: weights at W5t + z*Ci*SP, per-sample stride 8*Ci*SP."""
    from . import capi
    base = D.conv_fwd(N, thw, SP, SP, Ci, lddx, (1, 1, 1), (1, 1, 1), (0, 0, 0), thw, flags=capi.F_ACCUM if accum else 0, groups=N)
    base["wgstride"] = 8 * Ci * SP
    base["Ci_real"] = NSLOT
    return [(z, _sub(base, s, e)) for z, s, e in classes(thw)]


def wgrad_descs(N, thw, Ci, ldx, compact=False):
    """dW5[n][z][ci][slot] += sum over the class's positions of x[n][i][ci] * dcols[n][i][slot]: one launch per class with the
    N clip-passes as batched problems; gradient at dW5 + z*Ci*SP, per-sample stride 8*Ci*SP.
    compact: every class has a gradient buffer of its own, [n][ci][slot] (per-sample stride Ci*SP) -- the form whose K-slice images
    (pc_wgrad_desc.ws_slices) pc_tail6_wgrad_map_slices adds in slice order."""
    per = thw[0] * thw[1] * thw[2]
    out = []
    for z, s, e in classes(thw):
        d = D.wgrad(1, e, Ci, ldx, thw, SP, SP, (1, 1, 1), (1, 1, 1), (0, 0, 0))
        d.update(ioff0=list(s), Td=thw[0], Hd=thw[1], Wd=thw[2], doff=list(s),
                 nbatch=N, dbstride=per * ldx, sbstride=per * SP, gbstride=(1 if compact else 8) * Ci * SP, Cs_real=NSLOT)
        out.append((z, d))
    return out
