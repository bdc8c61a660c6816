"""Round 6: split-K weight gradients WITHOUT atomics (pc_wgrad_desc.ws_slices: every K slice leaves its partial sums as an image in a
workspace, the gradient re-layout adds the images in slice order), through the C-ABI -- autograd of every Unit3D / decoder conv under
loss.backward() (/root/reference/main_ucf101.py:183, models/pytorch_i3d.py:112-119).  Every kernel family that used fp32 atomics:
the stem's wgrad4_kernel, the row-segment kernels (fp32 and bf16-split), the generic split-K kernels (fp32 and bf16-split).  Bars:
the ordered sum agrees with the atomic one to fp32 rounding, equals an emulation of its own summation order bit for bit, and is
bit-identical from run to run -- also when the launch starts from an idle chip with cold caches, the pattern that exposed round 5's
missing LDS-DMA wait (tests/test_x6_gpu.py:295; VERDICT r5 #6: every LDS-DMA kernel gets that regression)."""
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from picons_amd import capi, desc, ops, spec
from tests.test_kernels_gpu import cl, w_oki

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _operands(Ci, Co, k, s, thw, N, seed=41):
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(N, Ci, *thw, generator=g) * torch.exp(0.5 * torch.randn(N, Ci, 1, 1, 1, generator=g)))
    pads = [spec.same_pad(thw[i], k[i], s[i]) for i in range(3)]
    othw = tuple((thw[i] + pads[i][0] + pads[i][1] - k[i]) // s[i] + 1 for i in range(3))
    dy = torch.randn(N, Co, *othw, generator=g)
    return x, dy, othw, [p[0] for p in pads]


def _cold(junk, it):
    """The launch that follows starts from an idle chip with cold L2 / Infinity Cache."""
    junk.fill_(float(it))
    torch.cuda.synchronize()
    time.sleep(0.002)


# (Ci, Co, k, stride, thw, N, flags, route): route as pc_wgrad_work reports it (0 stem, 1 row-segment, 3 generic split-K)
ORDERED_CASES = [
    (4, 64, (7, 7, 7), (2, 2, 2), (8, 56, 56), 2, capi.WG_CS3, 0),            # wgrad4_kernel, packed 3-channel columns
    (4, 64, (7, 7, 7), (2, 2, 2), (8, 56, 56), 2, capi.WG_CS3 | capi.WG_X6, 0),  # wgrad4_x6_kernel (two 16-position sub-chunks per chunk)
    (64, 64, (3, 3, 3), (1, 1, 1), (2, 8, 112), 2, 0, 1),                      # wgrad3_kernel, 56-position segments
    (64, 64, (3, 3, 3), (1, 1, 1), (2, 8, 112), 2, capi.WG_X6, 1),             # wgrad3_x6_kernel<64, 64, 64, 2>
    (64, 192, (3, 3, 3), (2, 1, 1), (4, 6, 56), 2, capi.WG_X6, 1),             # temporal stride 2
    (96, 128, (3, 3, 3), (1, 1, 1), (2, 5, 28), 2, capi.WG_X6, 1),             # wgrad3_x6_kernel<128, 32, 32, 4>
    (128, 256, (3, 3, 3), (1, 1, 1), (1, 28, 28), 4, capi.WG_X6, 1),           # T = 1: the outer temporal taps are trimmed and never written
    (256, 288, (1, 1, 1), (1, 1, 1), (1, 28, 28), 16, capi.WG_X6, 3),          # wgrad_x6_kernel<64, 128>
    (528, 128, (1, 1, 1), (1, 1, 1), (1, 28, 28), 8, capi.WG_X6, 3),           # wgrad_x6_kernel<128, 128>
    (64, 64, (1, 1, 1), (1, 1, 1), (4, 56, 56), 4, capi.WG_X6, 3),             # one tile, hundreds of K slices (Conv3d_2b): folded first
    (48, 136, (1, 3, 3), (1, 1, 1), (1, 20, 20), 4, 0, 3),                     # wgrad_kernel (fp32), ragged widths
    (64, 96, (1, 3, 3), (1, 1, 1), (2, 14, 30), 4, capi.WG_X6, 3),                # widths that are not multiples of 28
]


@pytest.mark.parametrize("Ci,Co,k,s,thw,N,flags,route", ORDERED_CASES)
def test_ordered_wgrad_matches_atomic_and_is_bit_identical(Ci, Co, k, s, thw, N, flags, route):
    x, dy, othw, pf = _operands(Ci, Co, k, s, thw, N)
    taps = int(np.prod(k))
    wd = desc.trim_wgrad(dict(desc.wgrad(N, othw, Co, Co, thw, Ci, Ci, k, s, pf), flags=flags))
    from picons_amd.plan import wgrad_work
    assert wgrad_work(wd)["route"] == route
    xg, dyg = cl(x), cl(dy)
    atomic = ops.conv_wgrad(wd, dyg, xg, torch.zeros(Co, taps, Ci, device=DEV))
    ns = ops.wgrad_slices(wd)
    image = Co * taps * Ci
    got, ws = ops.conv_wgrad_ordered(wd, dyg, xg)
    # the ordered sum is the same sum in another order
    err = ((got - atomic.view(-1)).norm() / atomic.norm()).item()
    assert err <= 2e-6, "ordered vs atomic split-K: rel-L2 %.3e over %d slices" % (err, ns)
    if flags & capi.WG_CS3:
        assert torch.all(ws.view(ns, Co, taps, Ci)[..., 3] == 0), "padding column of the slice images was written"
    # bit-identical reruns, from an idle chip with cold caches too (no atomics and every LDS-DMA tile waited for)
    junk = torch.empty(96 << 20, device=DEV)
    first = ws.clone()
    for it in range(12):
        _cold(junk, it)
        ops.conv_wgrad(dict(wd, ws_slices=ns), dyg, xg, ws)
        assert torch.equal(ws, first), "launch %d from an idle chip differs from the first" % it
    # the re-layout adds the images in slice order: kernel layout [Co][taps][Ci] -> the reference's [Co][Ci][taps] (+ fold for many slices)
    G = torch.full((Co, Ci, taps), float("nan"), device=DEV)
    nim, stride, imgs = ns, image, ws.clone()
    if ns > 128:
        grp = capi.lib().pc_wgrad_fold_group()
        ops.wgrad_fold(imgs, image, ns)
        emu = first.view(ns, image)
        sums = []
        for g0 in range(0, ns, grp):
            acc = emu[g0].clone()
            for q in range(g0 + 1, min(ns, g0 + grp)):
                acc += emu[q]
            sums.append(acc)
        nim, stride = len(sums), grp * image
        for q, t in enumerate(sums):
            assert torch.equal(imgs.view(ns, image)[q * grp], t), "fold of group %d is not the in-order sum" % q
    else:
        sums = list(first.view(ns, image))
    ops.transpose_multi([(imgs, G, Co, taps, Ci, taps * Ci, Ci, Ci * taps, taps, 0, nim, stride)])
    want = sums[0].clone()
    for t in sums[1:]:
        want += t
    assert torch.equal(G, want.view(Co, taps, Ci).permute(0, 2, 1)), "re-layout of %d images is not their in-order sum" % nim
    # and against fp64 autograd
    w = torch.zeros(Co, Ci, *k, dtype=torch.float64, requires_grad=True)
    pads = [spec.same_pad(thw[i], k[i], s[i]) for i in range(3)]
    xp = F.pad(x.double(), (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    F.conv3d(xp, w, None, s).backward(dy.double())
    ref = w.grad.reshape(Co, Ci, taps)
    if flags & capi.WG_CS3:
        ref = ref[:, :3]; G = G[:, :3]
    rel = ((G.cpu().double() - ref).norm() / ref.norm()).item()
    assert rel < 5e-6, rel


def test_ordered_wgrad_refuses_a_short_workspace_and_plain_stores():
    x, dy, othw, pf = _operands(64, 64, (3, 3, 3), (1, 1, 1), (2, 8, 56), 2)
    wd = desc.wgrad(2, othw, 64, 64, (2, 8, 56), 64, 64, (3, 3, 3), (1, 1, 1), pf)
    ns = ops.wgrad_slices(wd)
    assert ns > 1
    ws = torch.zeros(ns * 64 * 27 * 64, device=DEV)
    with pytest.raises(RuntimeError, match="slice images"):
        ops.conv_wgrad(dict(wd, ws_slices=ns - 1), cl(dy), cl(x), ws)
    with pytest.raises(RuntimeError, match="ws_slices"):
        ops.conv_wgrad(dict(wd, ws_slices=ns, splitk=-1), cl(dy), cl(x), ws)


# ---- launches from an idle chip: the LDS-DMA kernels the ordered weight-gradient cases above do not reach
def test_fp32_glds_conv_from_an_idle_chip_is_bit_identical():
    """conv_gemm_glds_kernel (fp32 MFMA, both tiles by LDS-DMA, two- and three-deep rings): a chip-filling launch and one of <= 256 blocks."""
    junk = torch.empty(96 << 20, device=DEV)
    for Ci, Co, k, thw, N in [(64, 128, (3, 3, 3), (4, 28, 28), 16), (192, 96, (1, 1, 1), (2, 28, 28), 8)]:
        x, _dy, othw, pf = _operands(Ci, Co, k, (1, 1, 1), thw, N, seed=43)
        g = torch.Generator().manual_seed(44)
        wk = w_oki(torch.randn(Co, Ci, *k, generator=g) / np.sqrt(Ci * np.prod(k)))
        d = desc.conv_fwd(N, thw, Ci, Ci, Co, Co, k, (1, 1, 1), pf, thw, act=capi.ACT_RELU)
        xg = cl(x)
        first = ops.conv_fwd(d, xg, wk, torch.empty(N, *thw, Co, device=DEV)).clone()
        out = torch.empty_like(first)
        for it in range(20):
            _cold(junk, it)
            ops.conv_fwd(d, xg, wk, out)
            assert torch.equal(out, first), "fp32 LDS-DMA conv, launch %d from an idle chip differs" % it


def test_wino_f2_conv_from_an_idle_chip_is_bit_identical():
    """wino_conv_kernel (F(2x2, 3x3): raw patch two chunks ahead, U one chunk ahead by LDS-DMA) at a 28^2 layer's shape."""
    N, T, H, W, Ci, Co = 8, 2, 28, 28, 128, 192
    g = torch.Generator().manual_seed(45)
    x = torch.relu(torch.randn(N, T, H, W, Ci, generator=g)).to(DEV)
    w = (torch.randn(Co, Ci, 3, 3, 3, generator=g) / np.sqrt(Ci * 27)).to(DEV).contiguous()
    U = ops.wino_weights(w, Co, Ci, 3)
    d = ops.wino_desc(N, T, H, W, Ci, Ci, Co, Co, 3, act=capi.ACT_NONE, flags=0)
    first = torch.empty(N, T, H, W, Co, device=DEV)
    ops.wino_conv(d, x, U, first)
    first = first.clone()
    out = torch.empty_like(first)
    junk = torch.empty(96 << 20, device=DEV)
    for it in range(20):
        _cold(junk, it)
        ops.wino_conv(d, x, U, out)
        assert torch.equal(out, first), "F(2x2, 3x3) launch %d from an idle chip differs" % it


# ---- the merged decoder tail's gradients (capsules_ucf101.py:504-509 under loss.backward()): the last four tensors that used fp32 atomics
def test_merged_tail_gradients_carry_no_atomics():
    """Per-class weight gradients into K-slice workspaces + pc_tail6_wgrad_map_slices, pc_tail6_bias_sums_ws, pc_tail_grads: equal to the atomic
    forms to fp32 rounding, equal to an emulation of their summation order bit for bit, and bit-identical from an idle chip."""
    from picons_amd import tail6
    g = torch.Generator().manual_seed(61)
    N, Ci, I = 4, 128, (3, 30, 28)
    SP = tail6.SP
    x = torch.relu(torch.randn(N, *I, Ci, generator=g)).to(DEV)
    dout = torch.randn(N, *[2 * v for v in I], generator=g).to(DEV)
    dcols = torch.empty(N, *I, SP, device=DEV)
    ops.tail6_scatter(dout, N, *I, dcols)
    # atomic form
    dW5 = torch.zeros(N, 8, Ci, SP, device=DEV)
    for z, d in tail6.wgrad_descs(N, I, Ci, Ci):
        ops.conv_wgrad(dict(d, flags=capi.WG_X6), x, dcols, dW5.view(-1)[z * Ci * SP:])
    Gc_atomic = ops.tail6_wgrad_map(dW5, N, Ci, torch.empty(N, Ci, 27, 32, device=DEV))
    # ordered form
    wds = [(z, dict(d, flags=capi.WG_X6)) for z, d in tail6.wgrad_descs(N, I, Ci, Ci, compact=True)]
    ns8 = [0] * 8
    for z, d in wds:
        ns8[z] = ops.wgrad_slices(d)
    assert max(ns8) > 1, ns8
    image = N * Ci * SP
    ws = torch.zeros(sum(ns8) * image, device=DEV)
    zoff = np.concatenate([[0], np.cumsum(ns8)[:-1]]) * image

    def run(junk=None, it=0):
        for z, d in wds:
            if junk is not None:
                _cold(junk, it)
            ops.conv_wgrad(dict(d, ws_slices=ns8[z]), x, dcols, ws[int(zoff[z]):])
        return ops.tail6_wgrad_map_slices(ws, ns8, N, Ci, torch.full((N, Ci, 27, 32), float("nan"), device=DEV))
    first_ws = None

    def images():          # the K-slice images as the launches left them (the map folds a class's images into its image 0)
        for z, d in wds:
            ops.conv_wgrad(dict(d, ws_slices=ns8[z]), x, dcols, ws[int(zoff[z]):])
        return ws.clone()
    first_ws = images()
    Gc = run()
    err = ((Gc - Gc_atomic).norm() / Gc_atomic.norm()).item()
    assert err <= 2e-6, err
    assert torch.all(Gc[..., 27:] == 0)
    # emulation of the order: a class's images in slice order, then the class sums in class order
    tap, j = np.meshgrid(np.arange(27), np.arange(27), indexing="ij")
    k4 = [tap // 9, (tap // 3) % 3, tap % 3]; ks = [j // 9, (j // 3) % 3, j % 3]
    slot = torch.from_numpy(((k4[0] + ks[0]) * 5 + k4[1] + ks[1]) * 5 + k4[2] + ks[2]).to(DEV)
    bad = [torch.from_numpy((k4[a] == 0) & (ks[a] == 2)).to(DEV) for a in range(3)]
    acc = torch.zeros(N, Ci, 27, 27, device=DEV)
    for z in range(8):
        skip = torch.zeros(27, 27, dtype=torch.bool, device=DEV)
        for a, bit in enumerate((4, 2, 1)):
            if z & bit:
                skip |= bad[a]
        if not ns8[z]:
            continue
        cls = first_ws[int(zoff[z]):int(zoff[z]) + image].clone()
        for k in range(1, ns8[z]):
            cls += first_ws[int(zoff[z]) + k * image:int(zoff[z]) + (k + 1) * image]
        assert torch.equal(ws[int(zoff[z]):int(zoff[z]) + image], cls), "class %d: image 0 is not the in-order sum of the class's slice images" % z
        acc = torch.where(skip, acc, acc + cls.view(N, Ci, SP)[:, :, slot])
    assert torch.equal(Gc[..., :27], acc), "pc_tail6_wgrad_map_slices is not the in-order sum of the class sums"
    first_Gc = Gc.clone()
    junk = torch.empty(96 << 20, device=DEV)
    for it in range(4):
        assert torch.equal(run(junk, it), first_Gc), "tail weight gradients, run %d differs" % it
    assert torch.equal(images(), first_ws), "the K-slice images differ from the first run's"
    # bias sums
    sums_atomic = ops.tail6_bias_sums(dout, N, *I, torch.empty(N, 32, device=DEV))
    part = torch.full((ops.tail6_bias_sums_ws_floats(N, *I),), float("nan"), device=DEV)        # needs no initialisation
    sums = ops.tail6_bias_sums(dout, N, *I, torch.full((N, 32), float("nan"), device=DEV), ws=part)
    assert torch.allclose(sums[:, :27], sums_atomic[:, :27], rtol=2e-5, atol=2e-4) and torch.all(sums[:, 27:] == 0)
    nb = part.numel() // (N * 32)
    emu = torch.zeros(N, 32, device=DEV)
    for b in range(nb):
        emu += torch.nan_to_num(part.view(N, nb, 32)[:, b])
    assert torch.equal(sums[:, :27], emu[:, :27])
    for it in range(4):
        _cold(junk, it)
        assert torch.equal(ops.tail6_bias_sums(dout, N, *I, torch.empty(N, 32, device=DEV), ws=part), sums)


def test_tail_grads_are_bit_identical_and_match_fp64():
    """pc_tail_grads_ws at the model's shape (N = 16 clip-passes, 128 -> 128 channels, 27 x 27 taps): the smooth-weight gradient's 256 block
    partials are stored and added in block order (no atomics since round 6); the workspace needs no initialisation."""
    g = torch.Generator().manual_seed(62)
    N, Ci, Co, taps, J = 16, 128, 128, 27, 27
    Gc = torch.randn(N, Ci, taps, 32, generator=g); Gc[..., 27:] = 0
    sums = torch.randn(N, 32, generator=g); sums[:, 27:] = 0
    W4 = torch.randn(Ci, Co, taps, generator=g) * 0.1
    b4 = torch.randn(Co, generator=g)
    cs = (torch.rand(N, Co, generator=g) < 0.5).float() * 2
    Wp = torch.randn(Co, J, generator=g) * 0.2
    dv = [t.to(DEV).contiguous() for t in (Gc, sums, W4, b4, cs, Wp)]
    outs = []
    junk = torch.empty(96 << 20, device=DEV)
    part = torch.full((int(capi.lib().pc_tail_grads_ws_floats(N, Ci, Co)),), float("nan"), device=DEV)
    for it in range(5):
        dW4 = torch.full((Ci, Co, taps), float("nan"), device=DEV); db4 = torch.full((Co,), float("nan"), device=DEV)
        dWp = torch.full((Co, J), float("nan"), device=DEV); dbp = torch.full((1,), float("nan"), device=DEV)
        _cold(junk, it)
        capi.call("pc_tail_grads_ws", *[ops.ptr(t) for t in dv], N, Ci, Co, taps, J, 13, ops.ptr(dW4), ops.ptr(db4), ops.ptr(dWp), ops.ptr(dbp), 0, ops.ptr(part),
                  ops.stream())
        outs.append([t.clone() for t in (dW4, db4, dWp, dbp)])
        for a, b in zip(outs[0], outs[-1]):
            assert torch.equal(a, b), "pc_tail_grads run %d differs" % it
    G64, W64, cs64, Wp64, s64, b64 = Gc[..., :27].double(), W4.double(), cs.double(), Wp.double(), sums[:, :27].double(), b4.double()
    ref_dWp = torch.einsum("nitj,iot,no->oj", G64, W64, cs64) + torch.einsum("o,no,nj->oj", b64, cs64, s64)
    ref_dW4 = torch.einsum("nitj,no,oj->iot", G64, cs64, Wp64)
    for name, got, ref in (("dWp", outs[0][2], ref_dWp), ("dW4", outs[0][0], ref_dW4)):
        rel = ((got.cpu().double() - ref).norm() / ref.norm()).item()
        assert rel < 3e-6, (name, rel)
    # accumulate form: dWp += the same
    dW4, db4, dWp, dbp = [t.clone() for t in outs[0]]
    capi.call("pc_tail_grads_ws", *[ops.ptr(t) for t in dv], N, Ci, Co, taps, J, 13, ops.ptr(dW4), ops.ptr(db4), ops.ptr(dWp), ops.ptr(dbp), 1, ops.ptr(part),
              ops.stream())
    assert torch.allclose(dWp, 2 * outs[0][2], rtol=1e-6, atol=1e-6)
    # the atomic form (no workspace) agrees to fp32 rounding
    dW4, db4, dWp, dbp = [torch.empty_like(t) for t in outs[0]]
    capi.call("pc_tail_grads", *[ops.ptr(t) for t in dv], N, Ci, Co, taps, J, 13, ops.ptr(dW4), ops.ptr(db4), ops.ptr(dWp), ops.ptr(dbp), 0, ops.stream())
    assert ((dWp - outs[0][2]).norm() / outs[0][2].norm()).item() < 2e-6 and torch.equal(dW4, outs[0][0])
