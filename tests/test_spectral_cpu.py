"""CPU: the host-side algebra of the two work-reducing forms, checked in float64 without any kernel:
 - spectral.matrices / matrices_full: row DFT -> three-multiplication products per frequency (one for DC / Nyquist)
   -> inverse DFT reproduces a 'valid' correlation (PrimaryCaps) resp. a full convolution (upsample1) along x;
 - tail6.classes tile the input lattice exactly once, and the 8-class / 125-slot merged tail reproduces
   upsample4 -> Dropout3d -> smooth (capsules_ucf101.py:504-509) including the cropped-position border rule."""
import itertools

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from picons_amd import spectral, tail6


def _planes_weights(w, tw, Ur):
    """w [KX] real taps -> per-plane scalars in the kernels' order (pc_wspec_fwd)."""
    U = tw.shape[0]
    out = []
    for u in range(U):
        wr = float((w * tw[u, :, 0]).sum()); wi = float((w * tw[u, :, 1]).sum())
        out += [wr] if u >= U - Ur else [wr, wr - wi, wr + wi]
    return np.array(out)


@pytest.mark.parametrize("P,KX", [(28, 9), (14, 9), (13, 3), (8, 5)])
def test_row_spectral_matrices_reproduce_valid_correlation(P, KX):
    rng = np.random.default_rng(P * 100 + KX)
    m = {k: v.astype(np.float64) for k, v in spectral.matrices(P, KX).items()}
    cplx, real = spectral.freq_order(P)
    assert m["F"].shape == (spectral.n_planes(P), P) and spectral.n_planes(P) == 3 * len(cplx) + len(real)
    x = rng.standard_normal(P); w = rng.standard_normal(KX)
    t = (m["F"] @ x) * _planes_weights(w, m["tw"], len(real))         # one real product per plane
    y = m["G"] @ t
    ref = np.array([sum(w[k] * x[o + k] for k in range(KX)) for o in range(P - KX + 1)])
    assert np.abs(y - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())      # float32 tables
    assert np.array_equal(m["Ft"], m["F"].T) and np.array_equal(m["Gt"], m["G"].T)


@pytest.mark.parametrize("W,KX", [(20, 9), (6, 9), (5, 3)])
def test_row_spectral_matrices_reproduce_full_convolution(W, KX):
    P = W + KX - 1
    rng = np.random.default_rng(W * 100 + KX)
    m = {k: v.astype(np.float64) for k, v in spectral.matrices_full(P, W, KX).items()}
    _c, real = spectral.freq_order(P)
    x = rng.standard_normal(W); w = rng.standard_normal(KX)
    y = m["G"] @ ((m["F"] @ x) * _planes_weights(w, m["tw"], len(real)))
    ref = np.convolve(x, w)                                             # ConvTranspose, stride 1: y[o] = sum_k x[o-k] w[k]
    assert y.shape == ref.shape and np.abs(y - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("thw", [(4, 112, 112), (2, 3, 5), (1, 4, 4), (2, 1, 3)])
def test_tail_position_classes_tile_the_lattice_once(thw):
    seen = np.zeros(thw, dtype=int)
    for z, s, e in tail6.classes(thw):
        assert all(((z >> (2 - d)) & 1) == (s[d] == 0) for d in range(3))
        seen[s[0]:s[0] + e[0], s[1]:s[1] + e[1], s[2]:s[2] + e[2]] += 1
    assert (seen == 1).all()


def test_merged_tail_algebra_matches_the_two_transposed_convs():
    torch.manual_seed(0)
    N, Ci, Co, I = 2, 3, 4, (2, 3, 4)
    f64 = dict(dtype=torch.float64)
    x = torch.randn(N, Ci, *I, **f64)
    W4 = torch.randn(Ci, Co, 3, 3, 3, **f64); b4 = torch.randn(Co, **f64)
    Ws = torch.randn(Co, 1, 3, 3, 3, **f64); bs = torch.randn(1, **f64)
    cs = (torch.rand(N, Co, **f64) < 0.5).double() * 2
    u4 = F.conv_transpose3d(x, W4, b4, stride=2, padding=1, output_padding=1) * cs.view(N, Co, 1, 1, 1)
    ref = F.conv_transpose3d(u4, Ws, bs, stride=1, padding=1)[:, 0]
    O = tuple(2 * i for i in I)
    Wc = torch.einsum("iokpq,no,oabc->nikpqabc", W4, cs, Ws[:, 0])        # combined weights [n][ci][k4][ks]
    bc = torch.einsum("o,no,oabc->nabc", b4, cs, Ws[:, 0])

    def pairs(k5, first):                                                  # csrc/tail6.hip npairs / pair_k4
        return [(k4, k5 - k4) for k4 in range(3) if 0 <= k5 - k4 <= 2 and not (first and k5 == 2 and k4 == 0)]
    cols = torch.zeros(N, *I, 5, 5, 5, **f64)
    for z, s, e in tail6.classes(I):
        first = [(z >> 2) & 1, (z >> 1) & 1, z & 1]
        W5 = torch.zeros(N, Ci, 5, 5, 5, **f64)
        for st, sh, sw in itertools.product(range(5), repeat=3):
            for (a, A), (b, B), (c, C) in itertools.product(pairs(st, first[0]), pairs(sh, first[1]), pairs(sw, first[2])):
                W5[:, :, st, sh, sw] += Wc[:, :, a, b, c, A, B, C]
        sl = tuple(slice(s[d], s[d] + e[d]) for d in range(3))
        cols[(slice(None),) + sl] = torch.einsum("nithw,nicde->nthwcde", x[(slice(None), slice(None)) + sl], W5)
    out = torch.zeros(N, *O, **f64)
    for o in itertools.product(*[range(v) for v in O]):
        acc = bs.expand(N).clone()
        for ks in itertools.product(range(3), repeat=3):                   # bias path: smooth tap ks reads o + 1 - ks
            if all(0 <= o[d] + 1 - ks[d] < O[d] for d in range(3)):
                acc = acc + bc[:, ks[0], ks[1], ks[2]]
        terms = [[((o[d] + 2 - k5) // 2, k5) for k5 in range(5) if (o[d] + 2 - k5) % 2 == 0 and 0 <= (o[d] + 2 - k5) // 2 < I[d]]
                 for d in range(3)]
        for (it, kt), (ih, kh), (iw, kw) in itertools.product(*terms):
            acc = acc + cols[:, it, ih, iw, kt, kh, kw]
        out[(slice(None),) + o] = acc
    assert (out - ref).abs().max().item() < 1e-10 * ref.abs().max().item()
