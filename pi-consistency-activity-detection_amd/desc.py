"""Descriptor builders for the gather-GEMM convolution family (include/picons.h pc_conv_desc /
pc_wgrad_desc).  Pure Python so the mapping of nn.Conv3d / nn.ConvTranspose3d and their
gradients onto the one kernel form can be unit-tested on CPU against torch (tests/test_desc.py).

A descriptor is a plain dict of ints / 3-lists with the struct's field names.
"""
from math import ceil


def _t3(v):
    return [int(v[0]), int(v[1]), int(v[2])]


def conv_out(size, k, s, pf, pb):
    return (size + pf + pb - k) // s + 1


def conv_fwd(N, in_thw, Ci, ldi, Co, ldo, k, stride, pad_front, out_thw, act=0, flags=0, groups=1, ldw=None):
    """nn.Conv3d forward: out[o] = sum_k in[o*s - pf + k] w[co][k][ci]."""
    return dict(N=N, Ti=in_thw[0], Hi=in_thw[1], Wi=in_thw[2], Ci=Ci, ldi=ldi,
                Tq=out_thw[0], Hq=out_thw[1], Wq=out_thw[2], To=out_thw[0], Ho=out_thw[1], Wo=out_thw[2],
                Co=Co, ldo=ldo, ostr=[1, 1, 1], ooff=[0, 0, 0], istr=_t3(stride), ntap=_t3(k),
                ioff0=[-int(p) for p in pad_front], istep=[1, 1, 1], wk0=[0, 0, 0], wkstep=[1, 1, 1],
                KT=k[0], KH=k[1], KW=k[2], ldw=ldw or Ci, act=act, flags=flags, wgstride=0, bgstride=0, act_c0=0, groups=groups)


def transposed_classes(N, small_thw, Cs, lds_, big_thw, Cb, ldb, k, stride, pad_front, act=0, flags=0, groups=1, ldw=None):
    """The 'scatter-adjoint' form shared by conv dgrad and ConvTranspose forward:
        big[o, cb] = sum_{i,k : o = i*s - pf + k} small[i, cs] * w[cb][k][cs]
    as one descriptor per output-parity class (prod(stride) of them): o = s*q + p,
    taps k = k0 + a*s with k0 = (p+pf) % s, input i = q + (p+pf-k0)/s - a."""
    out = []
    s = _t3(stride)
    for pt in range(s[0]):
        for ph in range(s[1]):
            for pw in range(s[2]):
                p = [pt, ph, pw]
                d = dict(N=N, Ti=small_thw[0], Hi=small_thw[1], Wi=small_thw[2], Ci=Cs, ldi=lds_,
                         To=big_thw[0], Ho=big_thw[1], Wo=big_thw[2], Co=Cb, ldo=ldb, ostr=s, ooff=p,
                         istr=[1, 1, 1], istep=[-1, -1, -1], KT=k[0], KH=k[1], KW=k[2], ldw=ldw or Cs,
                         act=act, flags=flags, wgstride=0, bgstride=0, act_c0=0, groups=groups)
                q, ntap, ioff0, wk0 = [], [], [], []
                empty = False
                for dim in range(3):
                    nq = ceil((big_thw[dim] - p[dim]) / s[dim]) if big_thw[dim] > p[dim] else 0
                    k0 = (p[dim] + pad_front[dim]) % s[dim]
                    nt = ceil((k[dim] - k0) / s[dim]) if k[dim] > k0 else 0
                    if nq == 0:
                        empty = True
                    if nt == 0:
                        raise ValueError("parity class without taps (kernel smaller than stride) is not supported")
                    q.append(nq); ntap.append(nt); wk0.append(k0)
                    ioff0.append((p[dim] + pad_front[dim] - k0) // s[dim])
                if empty:
                    continue
                d.update(Tq=q[0], Hq=q[1], Wq=q[2], ntap=ntap, ioff0=ioff0, wk0=wk0, wkstep=s)
                out.append(d)
    return out


def split_lattice(d, dim, starts):
    """Split a conv descriptor's output lattice along `dim` (0/1/2) at the given start indices -> list of
    descriptors covering [starts[i], starts[i+1]).  Used to align 128-row tiles with narrow spatial zones so the
    kernel's tile-level tap box is tight (PC_F_NFAST launches)."""
    key = ("Tq", "Hq", "Wq")[dim]
    total = d[key]
    bounds = [b for b in starts if b < total] + [total]
    out = []
    for q0, q1 in zip(bounds[:-1], bounds[1:]):
        if q1 <= q0:
            continue
        e = dict(d)
        e[key] = q1 - q0
        e["ooff"] = list(d["ooff"]); e["ioff0"] = list(d["ioff0"])
        e["ooff"][dim] += q0 * d["ostr"][dim]
        e["ioff0"][dim] += q0 * d["istr"][dim]
        out.append(e)
    return out


def wgrad(N, dense_thw, Cd, ldd, gath_thw, Cs, lds_, k, stride, pad_front, splitk=0):
    """g[cd][k][cs] += sum_{n,q} D[n,q,cd] * S[n, q*s - pf + k, cs]."""
    return dict(N=N, Tq=dense_thw[0], Hq=dense_thw[1], Wq=dense_thw[2], Cd=Cd, ldd=ldd,
                Ts=gath_thw[0], Hs=gath_thw[1], Ws=gath_thw[2], Cs=Cs, lds=lds_,
                istr=_t3(stride), ntap=_t3(k), ioff0=[-int(p) for p in pad_front], istep=[1, 1, 1],
                wk0=[0, 0, 0], KT=k[0], KH=k[1], KW=k[2], splitk=splitk, nbatch=0, dbstride=0, sbstride=0, gbstride=0, Td=0, Hd=0, Wd=0, doff=[0, 0, 0], flags=0, ws_slices=0)


def _valid_tap_range(Q, istr, ioff0, istep, ntap, I):
    """Taps a in [0, ntap) for which SOME lattice point q in [0,Q) reads inside [0,I):  contiguous [lo, hi]."""
    ok = []
    for a in range(ntap):
        base = ioff0 + a * istep
        lo_pos, hi_pos = base, (Q - 1) * istr + base            # istr >= 0
        ok.append(hi_pos >= 0 and lo_pos < I)
    if not any(ok):
        return 0, 0            # keep one (all-padding) tap so the kernel still writes zeros
    lo = ok.index(True)
    hi = len(ok) - 1 - ok[::-1].index(True)
    return lo, hi


def trim_conv(d):
    """Drop taps that can only ever gather zero padding (e.g. the two outer temporal taps of every 3x3x3 conv of
    Mixed_4b..4f, whose input has T = 1).  Exact: the result is unchanged, the K loop shrinks."""
    d = dict(d)
    I = (d["Ti"], d["Hi"], d["Wi"]); Q = (d["Tq"], d["Hq"], d["Wq"])
    nt, io, wk = list(d["ntap"]), list(d["ioff0"]), list(d["wk0"])
    for i in range(3):
        lo, hi = _valid_tap_range(Q[i], d["istr"][i], d["ioff0"][i], d["istep"][i], d["ntap"][i], I[i])
        nt[i] = hi - lo + 1
        io[i] = d["ioff0"][i] + lo * d["istep"][i]
        wk[i] = d["wk0"][i] + lo * d["wkstep"][i]
    d.update(ntap=nt, ioff0=io, wk0=wk)
    return d


def trim_wgrad(d):
    d = dict(d)
    I = (d["Ts"], d["Hs"], d["Ws"]); Q = (d["Tq"], d["Hq"], d["Wq"])
    nt, io, wk = list(d["ntap"]), list(d["ioff0"]), list(d["wk0"])
    for i in range(3):
        lo, hi = _valid_tap_range(Q[i], d["istr"][i], d["ioff0"][i], d["istep"][i], d["ntap"][i], I[i])
        nt[i] = hi - lo + 1
        io[i] = d["ioff0"][i] + lo * d["istep"][i]
        wk[i] = d["wk0"][i] + lo
    d.update(ntap=nt, ioff0=io, wk0=wk)
    return d


def pool(N, in_thw, C, ldi, out_thw, ldo, k, s, padf):
    return dict(N=N, Ti=in_thw[0], Hi=in_thw[1], Wi=in_thw[2], C=C, ldi=ldi,
                To=out_thw[0], Ho=out_thw[1], Wo=out_thw[2], ldo=ldo, k=_t3(k), s=_t3(s), padf=_t3(padf))


CONV_FIELDS = ["N", "Ti", "Hi", "Wi", "Ci", "ldi", "Tq", "Hq", "Wq", "To", "Ho", "Wo", "Co", "ldo",
               "ostr", "ooff", "istr", "ntap", "ioff0", "istep", "wk0", "wkstep", "KT", "KH", "KW", "ldw",
               "act", "flags", "wgstride", "bgstride", "act_c0", "groups"]
WGRAD_FIELDS = ["N", "Tq", "Hq", "Wq", "Cd", "ldd", "Ts", "Hs", "Ws", "Cs", "lds", "istr", "ntap", "ioff0",
                "istep", "wk0", "KT", "KH", "KW", "splitk", "nbatch", "dbstride", "sbstride", "gbstride", "Td", "Hd", "Wd", "doff", "flags", "ws_slices"]
POOL_FIELDS = ["N", "Ti", "Hi", "Wi", "C", "ldi", "To", "Ho", "Wo", "ldo", "k", "s", "padf"]


CONV_VEC3 = ("ostr", "ooff", "istr", "ntap", "ioff0", "istep", "wk0", "wkstep")


def unflatten_conv(flat):
    """Inverse of flatten(d, CONV_FIELDS): the int list an OP_CONV / OP_CONV_X6 op carries -> descriptor dict."""
    d, q = {}, 0
    for f in CONV_FIELDS:
        if f in CONV_VEC3:
            d[f] = [int(x) for x in flat[q:q + 3]]
            q += 3
        else:
            d[f] = int(flat[q])
            q += 1
    return d


def flatten(d, fields):
    """dict -> flat list of ints in struct order (what pc_op.i carries)."""
    out = []
    for f in fields:
        v = d[f]
        out.extend(int(x) for x in v) if isinstance(v, (list, tuple)) else out.append(int(v))
    return out
