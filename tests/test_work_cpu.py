"""CPU: the host-side work accounting behind bench.py's roofline numerator (pc_conv_work / pc_wgrad_work, no GPU call).
`valid` is checked against an independent count (a torch convolution of ones), `executed` against the closed form for
launches whose tiles lie inside one t-slice, and the orderings issued >= executed >= valid always."""
import numpy as np
import torch
import torch.nn.functional as F

from picons_amd import capi, desc as D, step as pstep
from picons_amd.plan import Plan, conv_work, wgrad_work, _conv_flops, _wgrad_flops


def _valid_by_torch(N, thw, Ci, Co, k, stride, pad):
    x = torch.ones(1, 1, *thw)
    w = torch.ones(1, 1, *k)
    return float(F.conv3d(x, w, stride=stride, padding=pad).sum()) * N * Ci * Co


def test_valid_macs_match_a_convolution_of_ones():
    for N, thw, Ci, Co, k, s in [(4, (4, 28, 28), 64, 64, (3, 3, 3), (1, 1, 1)), (2, (2, 14, 14), 32, 96, (3, 3, 3), (1, 1, 1)),
                                 (2, (1, 28, 28), 160, 320, (1, 3, 3), (1, 1, 1)), (2, (8, 56, 56), 4, 64, (7, 7, 7), (2, 2, 2))]:
        pad = tuple(x // 2 for x in k)
        othw = tuple((thw[i] + 2 * pad[i] - k[i]) // s[i] + 1 for i in range(3))
        d = D.trim_conv(D.conv_fwd(N, thw, Ci, Ci, Co, Co, k, s, pad, othw, groups=2))
        w = conv_work(d)
        want = _valid_by_torch(N, thw, Ci, Co, k, s, pad)
        assert w["valid"] == want, (thw, k, w, want)
        assert w["issued"] >= w["executed"] >= w["valid"] > 0
        assert 2 * w["executed"] <= _conv_flops(d)
        wd = D.trim_wgrad(D.wgrad(N, othw, Co, Co, thw, Ci, Ci, k, s, pad))
        ww = wgrad_work(wd)
        assert ww["valid"] == want and ww["issued"] >= ww["executed"] >= ww["valid"]
        assert 2 * ww["executed"] <= _wgrad_flops(wd)


def test_tap_box_of_tiles_inside_one_t_slice():
    """conv112 (64 -> 64, 3x3x3, 4x112x112, 16 clip-passes in 2 groups): 128-row tiles never straddle a t-slice (112*112 = 98 * 128)
    and always span two image rows, so each tile walks 2 of 3 kt taps at t = 0, 3 and all 27 taps otherwise: 10/12 of the descriptor."""
    d = D.trim_conv(D.conv_fwd(16, (4, 112, 112), 64, 64, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (4, 112, 112), groups=2))
    w = conv_work(d)
    assert (w["bm"], w["bn"], w["glds"]) == (128, 64, True) and w["blocks"] == 6272
    assert 2 * w["executed"] * 12 == _conv_flops(d) * 10
    assert w["issued"] == w["executed"]                       # no row / column padding on this shape
    # T = 2 (Conv3d_2c): every tile sees 2 of 3 kt taps
    d = D.trim_conv(D.conv_fwd(16, (2, 56, 56), 64, 64, 192, 192, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 56, 56), groups=2))
    w = conv_work(d)
    assert abs(2 * w["executed"] / _conv_flops(d) - 4.0 / 6.0) < 0.02
    # no padding at all: executed == valid == descriptor
    d = D.trim_conv(D.conv_fwd(16, (1, 28, 28), 832, 832, 256, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 28, 28), groups=2))
    w = conv_work(d)
    assert 2 * w["executed"] == 2 * w["valid"] == _conv_flops(d)


def test_stem_counts_three_real_channels():
    """The RGB stem: Ci = 4 in the kernel (16-byte pieces), 3 real; with PC_F_CI3 the padding channel's MFMA is not issued."""
    d = D.conv_fwd(16, (8, 224, 224), 4, 4, 64, 64, (7, 7, 7), (2, 2, 2), (2, 2, 2), (4, 112, 112), groups=2)     # SAME: pad 5 -> front 2
    d = D.trim_conv(d)
    d["Ci_real"] = 3
    d["flags"] = capi.F_CI3
    w3 = conv_work(d)
    d4 = dict(d, flags=0)
    w4 = conv_work(d4)
    assert w3["executed"] == w4["executed"] and w3["issued"] * 4 == w4["issued"] * 3
    assert w3["issued"] >= w3["executed"] and w3["executed"] * 4 <= _conv_flops(dict(d, Ci_real=4)) * 3 / 2 + 1


def test_step_totals_are_consistent():
    args = pstep.default_args(bv=True, n_frames=5, wt_cons=0.1)
    p = Plan(24, 112, n=2, groups=2, lanes=1)
    p.build_forward(); p.build_loss(args); p.build_backward(); p.build_adam()
    fe, fw = p.conv_flops_executed(), p.wgrad_flops_executed()
    fc, fg = p.flops(capi.OP_CONV), p.flops(capi.OP_WGRAD)
    for n in ("fwd", "bwd"):
        assert fe[n]["mfma"] >= fe[n]["executed"] >= fe[n]["valid"] and fe[n]["executed"] <= fc[n]
        assert fw[n]["mfma"] >= fw[n]["executed"] >= fw[n]["valid"] and fw[n]["executed"] <= fg[n]
    assert fe["fwd"]["executed"] > 0 and fw["bwd"]["executed"] > 0
    # every conv / wgrad op carries its work record (tools/launch_table.py pairs them with the trace)
    p.finalize()
    ops = [op for n in p.lists for op in p.lists[n] if op[0] in (capi.OP_CONV, capi.OP_WGRAD)]
    assert ops and all(id(op[1]) in p.op_work for op in ops)


def test_winograd_layers_in_the_plan_and_their_accounting(monkeypatch):
    """Which layers run in Winograd form at the bench size, what pc_wino_work books for them (16 transform-domain multiply-accumulates
    per 2x2 tile, tap and channel pair: 4/9 of the direct form's valid count away from the borders), and PICONS_WINO=0 restoring the
    gather-GEMM ops."""
    import ctypes as C
    from picons_amd import ops
    args = pstep.default_args(bv=True, n_frames=5, wt_cons=0.1)

    def build():
        p = Plan(24, 224, n=2, groups=2, lanes=4)
        p.build_forward(); p.build_loss(args); p.build_backward(); p.build_adam(); p.finalize()
        return p
    p = build()
    wino_layers = sorted(n for n, v in p.kw.items() if "wino_fwd" in v)
    assert wino_layers == sorted(["conv112.weight", "conv56.weight", "conv1.Conv3d_2c_3x3.conv3d.weight", "conv1.Mixed_3b.b1b.conv3d.weight",
                                  "conv1.Mixed_3c.b1b.conv3d.weight", "conv1.Mixed_3c.b2b.conv3d.weight"] +
                                 # one frame at 28x28: the b1b branch of every module, the b2b branches with >= 32 input channels
                                 ["conv1.Mixed_4%s.b1b.conv3d.weight" % m for m in "bcdef"] + ["conv1.Mixed_4e.b2b.conv3d.weight", "conv1.Mixed_4f.b2b.conv3d.weight"])
    count = lambda q, kind: sum(1 for n in q.lists for op in q.lists[n] if op[0] == kind)
    assert count(p, capi.OP_WINO_CONV) == 26 and count(p, capi.OP_WINO_WEIGHTS) == 26      # forward + input gradient each
    fz = p.wino_flops_executed()
    ex = sum(v["executed"] for v in fz.values()); mf = sum(v["mfma"] for v in fz.values())
    ref = sum(p.flops_reference_counted_wino().values())
    # 2.25x fewer than the direct form with F(2x2, 3x3) (more where temporal taps fall outside: 6.75x at one frame), 4x with F(4x4, 3x3), which
    # the 112 x 112 and 56 x 56 layers run in (Plan.wino_m)
    assert 0 < ex <= mf and 3.0 < ref / ex < 6.0
    wops = [(n, op[1]) for n in p.lists for op in p.lists[n] if op[0] == capi.OP_WINO_CONV]
    assert {i[2] for _n, i in wops if i[15] == 4} == {112, 56} and {i[2] for _n, i in wops if i[15] != 4} == {28, 56}
    # ... except ONE launch: the forward of Conv3d_2c (56 x 56, 64 -> 192, BatchNorm partial sums), the only one of the six that sits in the
    # trunk's forward, in front of EM routing (Plan.unit3d; PICONS_WINO4_TRUNK_FWD=1 would move it)
    f23_56 = [(n, i) for n, i in wops if i[15] != 4 and i[2] == 56]
    assert len(f23_56) == 1 and f23_56[0][0] == "fwd" and (f23_56[0][1][4], f23_56[0][1][6]) == (64, 192) and f23_56[0][1][10] & capi.F_BNPART
    assert sorted(op[1][4] for n in p.lists for op in p.lists[n] if op[0] == capi.OP_WINO_WEIGHTS).count(4) == 5
    monkeypatch.setenv("PICONS_WINO4", "0")
    q4 = build()
    fz2 = q4.wino_flops_executed()
    assert 1.9 < ref / sum(v["executed"] for v in fz2.values()) < 4.0
    assert all(op[1][15] in (0, 2) for n in q4.lists for op in q4.lists[n] if op[0] == capi.OP_WINO_CONV)
    monkeypatch.delenv("PICONS_WINO4")
    monkeypatch.setenv("PICONS_WINO_T1", "0")
    assert count(build(), capi.OP_WINO_CONV) == 12
    monkeypatch.delenv("PICONS_WINO_T1")
    # conv112 at bs = 8: 3136 blocks of 64 tiles x 64 channels, 10 of 12 temporal taps valid
    d = ops.wino_desc(16, 4, 112, 112, 64, 64, 64, 64, 3)
    out = (C.c_double * 3)()
    capi.check(capi.lib().pc_wino_work(C.byref(d), out))
    assert out[2] == 3136 and out[0] == out[1] == 16 * 10 * 16.0 * 56 * 56 * 64 * 64
    assert capi.lib().pc_wino_u_floats(64, 64, 3) == 3 * 1 * 8 * 8192 and capi.lib().pc_wino_u_floats(96, 64, 3) == 3 * 2 * 8 * 8192
    assert capi.lib().pc_wino_bnpart_rows(C.byref(d)) == 16 * 4 * 49 * 2
    # the F(4x4, 3x3) form of the same layer: 28 blocks of 28 (of 32) tiles x 64 channels per frame, 36 transform positions
    d4 = ops.wino_desc(16, 4, 112, 112, 64, 64, 64, 64, 3, m=4)
    capi.check(capi.lib().pc_wino_work(C.byref(d4), out))
    assert out[2] == 16 * 4 * 28 and out[1] == 16 * 10 * 36.0 * 28 * 28 * 64 * 64 and out[0] == out[1] * 32 / 28
    assert capi.lib().pc_wino4_u_floats(64, 64, 3) == 3 * 1 * 16 * 9216 and capi.lib().pc_wino_bnpart_rows(C.byref(d4)) == 16 * 4 * 28 * 2
    bad4 = ops.wino_desc(2, 2, 14, 16, 64, 64, 64, 64, 3, m=4)
    assert capi.lib().pc_wino_work(C.byref(bad4), out) != 0 and b"multiples of 4" in capi.lib().pc_last_error()
    bad4.m = 3
    assert capi.lib().pc_wino_work(C.byref(bad4), out) != 0 and b"m must be" in capi.lib().pc_last_error()
    # odd sizes and thin channel counts are refused by the host checks (no GPU call)
    bad = ops.wino_desc(2, 2, 15, 16, 64, 64, 64, 64, 3)
    assert capi.lib().pc_wino_work(C.byref(bad), out) != 0 and b"even" in capi.lib().pc_last_error()
    monkeypatch.setenv("PICONS_WINO", "0")
    q = build()
    assert count(q, capi.OP_WINO_CONV) == 0 and count(q, capi.OP_CONV) > count(p, capi.OP_CONV)
    assert sum(v["executed"] for v in q.conv_flops_executed().values()) > sum(v["executed"] for v in p.conv_flops_executed().values())


def test_every_bf16_split_conv_op_in_the_plan_has_the_shape_the_kernel_is_gated_on():
    """at the bench size (bs 8, 224^2, four lanes) no OP_CONV_X6 op may carry a channel count the LDS-DMA tiles cannot take
    (Ci % 32 != 0) or fewer than 150 blocks of its tile (measured slower than the fp32 kernel there), and every one is accepted by the
    library's own gate; the kernel-level numerics gates of tests/test_x6_gpu.py only cover shapes that satisfy this."""
    import ctypes as C
    from picons_amd.plan import _cdesc
    args = pstep.default_args(bv=True, n_frames=5, wt_cons=0.1)
    p = Plan(24, 224, n=8, groups=2, lanes=4)
    p.build_forward(); p.build_loss(args); p.build_backward(); p.build_adam(); p.finalize()
    x6 = [op for n in p.lists for op in p.lists[n] if op[0] == capi.OP_CONV_X6]
    assert len(x6) >= 25
    lib = capi.lib()
    for op in x6:
        d = D.unflatten_conv(op[1])
        assert D.flatten(d, D.CONV_FIELDS) == [int(v) for v in op[1]]
        w = p.op_work[id(op[1])]
        assert d["flags"] & capi.F_X6 and d["Ci"] % 32 == 0 and d["Ci"] >= 32, d
        assert w["blocks"] >= 150, (d, w["blocks"])
        assert lib.pc_conv_x6_ok(_cdesc(d)) == 1
    # and nothing that the gate would accept was left on the fp32 kernel by accident in the BACKWARD (the forward's trunk convs stay fp32 on
    # purpose: DESIGN.md 4, the numerics gate)
    asx6 = lambda op: dict(D.unflatten_conv(op[1]), flags=D.unflatten_conv(op[1])["flags"] | capi.F_X6)
    left = [op for op in p.lists["bwd"] if op[0] == capi.OP_CONV and lib.pc_conv_x6_ok(_cdesc(asx6(op))) == 1]
    assert not left, "%d backward conv ops pass pc_conv_x6_ok but run on the fp32 kernel" % len(left)
