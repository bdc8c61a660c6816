"""CPU: the descriptor builders map Conv3d / ConvTranspose3d and their gradients onto the one
gather-GEMM form correctly (checked with a numpy interpreter of the descriptor semantics)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from picons_amd import desc, spec
from tests.desc_interp import run_conv, run_wgrad


def cl(x):      # NCDHW torch -> NDHWC numpy float64
    return x.detach().permute(0, 2, 3, 4, 1).contiguous().double().numpy()


CASES = [  # Ci, Co, k, s, in thw, SAME?
    (4, 6, (3, 3, 3), (2, 1, 1), (4, 5, 6)),
    (4, 5, (7, 7, 7), (2, 2, 2), (6, 9, 8)),
    (8, 4, (1, 1, 1), (1, 1, 1), (2, 3, 3)),
    (4, 4, (1, 3, 3), (1, 1, 1), (1, 5, 5)),
]


@pytest.mark.parametrize("Ci,Co,k,s,thw", CASES)
def test_conv_same_fwd_dgrad_wgrad(Ci, Co, k, s, thw):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, Ci, *thw, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, Ci, *k, generator=g, dtype=torch.float64, requires_grad=True)
    pads = [spec.same_pad(thw[i], k[i], s[i]) for i in range(3)]
    xp = F.pad(x, (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    y = F.conv3d(xp, w, None, s)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    othw = tuple(y.shape[2:])
    pf = [p[0] for p in pads]
    taps = k[0] * k[1] * k[2]
    w_oki = w.detach().reshape(Co, Ci, taps).permute(0, 2, 1).contiguous().numpy()       # [O][taps][I]
    w_iko = w.detach().reshape(Co, Ci, taps).permute(1, 2, 0).contiguous().numpy()       # [I][taps][O]
    d = desc.conv_fwd(2, thw, Ci, Ci, Co, Co, k, s, pf, othw)
    np.testing.assert_allclose(run_conv(d, cl(x), w_oki), cl(y), atol=1e-10)
    dx = np.zeros((2,) + thw + (Ci,))
    for dd in desc.transposed_classes(2, othw, Co, Co, thw, Ci, Ci, k, s, pf):
        run_conv(dd, cl(dy), w_iko, out=dx)
    np.testing.assert_allclose(dx, cl(x.grad), atol=1e-10)
    gw = run_wgrad(desc.wgrad(2, othw, Co, Co, thw, Ci, Ci, k, s, pf), cl(dy), cl(x))    # [O][taps][I]
    np.testing.assert_allclose(gw, w.grad.reshape(Co, Ci, taps).permute(0, 2, 1).numpy(), atol=1e-9)


TCASES = [  # Ci, Co, k, s, pad, outpad, in thw
    (4, 6, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), (2, 3, 4)),
    (4, 3, (3, 3, 3), (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 3)),
    (8, 4, (1, 5, 5), (1, 1, 1), (0, 0, 0), (0, 0, 0), (1, 4, 4)),
]


@pytest.mark.parametrize("Ci,Co,k,s,pd,op,thw", TCASES)
def test_conv_transpose_fwd_dgrad_wgrad(Ci, Co, k, s, pd, op, thw):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, Ci, *thw, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Ci, Co, *k, generator=g, dtype=torch.float64, requires_grad=True)     # IO(T)HW
    y = F.conv_transpose3d(x, w, None, s, pd, op)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    othw = tuple(y.shape[2:])
    taps = k[0] * k[1] * k[2]
    w_oki = w.detach().reshape(Ci, Co, taps).permute(1, 2, 0).contiguous().numpy()       # [O][taps][I]
    w_iko = w.detach().reshape(Ci, Co, taps).permute(0, 2, 1).contiguous().numpy()       # [I][taps][O]
    out = np.zeros((2,) + othw + (Co,))
    for dd in desc.transposed_classes(2, thw, Ci, Ci, othw, Co, Co, k, s, pd):
        run_conv(dd, cl(x), w_oki, out=out)
    np.testing.assert_allclose(out, cl(y), atol=1e-10)
    d = desc.conv_fwd(2, othw, Co, Co, Ci, Ci, k, s, pd, thw)          # dgrad of ConvTranspose = strided conv
    np.testing.assert_allclose(run_conv(d, cl(dy), w_iko), cl(x.grad), atol=1e-10)
    gw = run_wgrad(desc.wgrad(2, thw, Ci, Ci, othw, Co, Co, k, s, pd), cl(x), cl(dy))    # [I][taps][O]
    np.testing.assert_allclose(gw, w.grad.reshape(Ci, Co, taps).permute(0, 2, 1).numpy(), atol=1e-9)


@pytest.mark.parametrize("thw,k,s", [((1, 5, 6), (3, 3, 3), (1, 1, 1)), ((2, 4, 4), (3, 3, 3), (2, 1, 1)), ((1, 1, 7), (3, 3, 3), (1, 1, 1))])
def test_tap_trimming_is_exact(thw, k, s):
    """Taps that can only read zero padding are dropped from the descriptors (T = 1 inputs of Mixed_4b..4f):
    same result, shorter K loop; wgrad leaves the dropped taps' gradient at zero."""
    g = torch.Generator().manual_seed(3)
    Ci, Co = 4, 5
    x = torch.randn(2, Ci, *thw, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, Ci, *k, generator=g, dtype=torch.float64, requires_grad=True)
    pads = [spec.same_pad(thw[i], k[i], s[i]) for i in range(3)]
    y = F.conv3d(F.pad(x, (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1])), w, None, s)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    othw = tuple(y.shape[2:]); pf = [p[0] for p in pads]; taps = k[0] * k[1] * k[2]
    w_oki = w.detach().reshape(Co, Ci, taps).permute(0, 2, 1).contiguous().numpy()
    w_iko = w.detach().reshape(Co, Ci, taps).permute(1, 2, 0).contiguous().numpy()
    d0 = desc.conv_fwd(2, thw, Ci, Ci, Co, Co, k, s, pf, othw)
    d = desc.trim_conv(d0)
    assert np.prod(d["ntap"]) < np.prod(d0["ntap"])
    np.testing.assert_allclose(run_conv(d, cl(x), w_oki), cl(y), atol=1e-10)
    dx = np.zeros((2,) + thw + (Ci,))
    for dd in desc.transposed_classes(2, othw, Co, Co, thw, Ci, Ci, k, s, pf):
        run_conv(desc.trim_conv(dd), cl(dy), w_iko, out=dx)
    np.testing.assert_allclose(dx, cl(x.grad), atol=1e-10)
    wd = desc.trim_wgrad(desc.wgrad(2, othw, Co, Co, thw, Ci, Ci, k, s, pf))
    assert np.prod(wd["ntap"]) < taps
    np.testing.assert_allclose(run_wgrad(wd, cl(dy), cl(x)), w.grad.reshape(Co, Ci, taps).permute(0, 2, 1).numpy(), atol=1e-9)


def test_split_lattice_is_exact():
    g = torch.Generator().manual_seed(4)
    Ci, Co, k, thw = 4, 3, (1, 5, 5), (1, 4, 6)
    x = torch.randn(2, Ci, *thw, generator=g, dtype=torch.float64)
    w = torch.randn(Ci, Co, *k, generator=g, dtype=torch.float64)
    y = F.conv_transpose3d(x, w)
    othw = tuple(y.shape[2:]); taps = 25
    w_oki = w.reshape(Ci, Co, taps).permute(1, 2, 0).contiguous().numpy()
    out = np.zeros((2,) + othw + (Co,))
    n = 0
    for dd in desc.transposed_classes(2, thw, Ci, Ci, othw, Co, Co, k, (1, 1, 1), (0, 0, 0)):
        for dz in desc.split_lattice(dd, 2, [0, 4, 8]):
            run_conv(desc.trim_conv(dz), cl(x), w_oki, out=out); n += 1
    assert n == 3
    np.testing.assert_allclose(out, cl(y), atol=1e-10)
