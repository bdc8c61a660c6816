"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz, produced
by tools/make_goldens.py in the authoring container).  CPU only."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import caps, i3d, losses, step as ostep
from picons_amd import spec, synthetic

T = lambda a: torch.from_numpy(np.asarray(a))


def close(a, b, atol, rtol=0.0, what=""):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a.astype(np.float64) - b.astype(np.float64))
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), "%s: max err %.3e (tol %.1e)" % (what, err.max(), atol)


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "stages.npz"))


@pytest.mark.parametrize("tag", ["u333", "u777", "u111"])
def test_unit3d(G, tag):
    P = {"u.conv3d.weight": T(G[tag + "_w"]).requires_grad_(True),
         "u.bn.weight": T(G[tag + "_gamma"]).requires_grad_(True),
         "u.bn.bias": T(G[tag + "_beta"]).requires_grad_(True),
         "u.bn.running_mean": torch.zeros(G[tag + "_w"].shape[0]),
         "u.bn.running_var": torch.ones(G[tag + "_w"].shape[0])}
    x = T(G[tag + "_x"]).requires_grad_(True)
    s = tuple(int(v) for v in G[tag + "_s"])
    y = i3d.unit3d(P, "u", x, s, True)
    y.backward(T(G[tag + "_dy"]))
    close(y, G[tag + "_y"], 2e-5, what="y")
    close(x.grad, G[tag + "_dx"], 5e-5, 1e-4, what="dx")
    close(P["u.conv3d.weight"].grad, G[tag + "_dw"], 2e-4, 1e-4, what="dw")
    close(P["u.bn.weight"].grad, G[tag + "_dgamma"], 2e-4, 1e-4, what="dgamma")
    close(P["u.bn.bias"].grad, G[tag + "_dbeta"], 2e-4, 1e-4, what="dbeta")
    close(P["u.bn.running_mean"], G[tag + "_rm"], 1e-6, what="rm")
    close(P["u.bn.running_var"], G[tag + "_rv"], 1e-6, what="rv")
    close(i3d.unit3d(P, "u", x.detach(), s, False), G[tag + "_y_eval"], 2e-5, what="y_eval")


@pytest.mark.parametrize("tag", ["p133", "p333s2", "p333s1", "p133odd"])
def test_maxpool_same(G, tag):
    x = T(G[tag + "_x"]).requires_grad_(True)
    k = tuple(int(v) for v in G[tag + "_k"]); s = tuple(int(v) for v in G[tag + "_s"])
    y = i3d.maxpool_same(x, k, s)
    y.backward(T(G[tag + "_dy"]))
    close(y, G[tag + "_y"], 0, what="y")
    close(x.grad, G[tag + "_dx"], 1e-6, what="dx")


def test_inception(G):
    P = {}
    for k in G.files:
        if k.startswith("inc_p_"):
            P["m." + k[len("inc_p_"):]] = T(G[k])
    for br, co in zip(["b0", "b1a", "b1b", "b2a", "b2b", "b3b"], [8, 8, 12, 4, 8, 8]):
        P["m.%s.bn.running_mean" % br] = torch.zeros(co)
        P["m.%s.bn.running_var" % br] = torch.ones(co)
    y = i3d.inception(P, "m", T(G["inc_x"]), [int(v) for v in G["inc_oc"]], True)
    close(y, G["inc_y"], 5e-5, what="inception")


@pytest.mark.parametrize("tag", ["capS", "capF"])
def test_capsule_head(G, tag):
    P = {"primary_caps.pose.weight": T(G[tag + "_pose_w"]), "primary_caps.pose.bias": T(G[tag + "_pose_b"]),
         "primary_caps.a.weight": T(G[tag + "_a_w"]), "primary_caps.a.bias": T(G[tag + "_a_b"]),
         "conv_caps.weights": T(G[tag + "_W"]).requires_grad_(True),
         "conv_caps.beta_u": T(G[tag + "_beta_u"]).requires_grad_(True),
         "conv_caps.beta_a": T(G[tag + "_beta_a"]).requires_grad_(True)}
    x = T(G[tag + "_x"]).requires_grad_(True)
    pc = caps.primary_caps(P, x)
    close(pc, G[tag + "_pc"], 2e-5, what="primary caps")
    B = G[tag + "_W"].shape[1]
    # conv_caps() reads spec.IN_CAPS; run the routing pieces directly for the small case
    b, h, w, _ = pc.shape
    pose = pc[..., :B * 16].reshape(b * h * w, B, 16)
    a_in = pc[..., B * 16:].reshape(b * h * w, B, 1)
    v = caps.votes(pose, P["conv_caps.weights"])
    mu, a_out = caps.em_routing(v, a_in, P["conv_caps.beta_u"], P["conv_caps.beta_a"])
    C = G[tag + "_W"].shape[2]
    out = torch.cat([mu.reshape(b, h, w, C * 16), a_out.reshape(b, h, w, C)], dim=3)
    close(out, G[tag + "_out"], 5e-5, what="conv caps out")
    out.backward(T(G[tag + "_dout"]))
    close(x.grad, G[tag + "_dx"], 1e-4, 2e-3, what="dx")
    close(P["conv_caps.weights"].grad, G[tag + "_dW"], 1e-4, 2e-3, what="dW")
    close(P["conv_caps.beta_u"].grad, G[tag + "_dbeta_u"], 1e-4, 2e-3, what="dbeta_u")
    close(P["conv_caps.beta_a"].grad, G[tag + "_dbeta_a"], 1e-6, 2e-3, what="dbeta_a")


def test_losses(G):
    x = T(G["spread_x"]).requires_grad_(True)
    l, al = losses.spread_loss(x, T(G["spread_t"]))
    l.backward()
    close(l, G["spread_loss"], 1e-7); close(al, G["spread_abs"], 1e-6); close(x.grad, G["spread_dx"], 1e-7)
    lg = T(G["seg_logits"]).requires_grad_(True)
    d = losses.dice_loss(lg, T(G["seg_t"])); b = losses.bce_logits(lg, T(G["seg_t"]))
    (d + b).backward()
    close(d, G["dice"], 1e-6); close(b, G["bce"], 1e-6); close(lg.grad, G["seg_dlogits"], 1e-8, 1e-5)
    close(losses.weighted_mse(T(G["wm_a"]), T(G["wm_b"]), T(G["wm_w5"])), G["wm_l5"], 1e-6)
    close(losses.weighted_mse(T(G["wm_a"]), T(G["wm_b"]), T(G["wm_w4"])), G["wm_l4"], 1e-6)
    r = np.array([losses.exp_rampup(100)(e) for e in (0, 1, 11, 50, 99, 100, 150)])
    close(r, G["ramp_100"], 0)


def _mask_inputs():
    g = np.random.default_rng(23)
    pred = g.normal(0, 2, (2, 1, 8, 224, 224)).astype(np.float32)
    flip = (pred[:, :, ::-1] * 0.7 + g.normal(0, 1, pred.shape)).astype(np.float32)
    return T(pred), T(flip)


def _check_mask(M, tag, m):
    m = m.numpy()
    assert list(m.shape) == list(M[tag + "_shape"]) and str(m.dtype) == str(M[tag + "_dtype"])
    close(m[..., ::7, ::7], M[tag + "_sample"], 1e-6, what=tag)
    close(m.sum(axis=(-1, -2)), M[tag + "_sum"], 1e-3, what=tag + " sum")
    close((m * m).sum(axis=(-1, -2)), M[tag + "_sumsq"], 1e-3, what=tag + " sumsq")


def test_masks(golden_dir):
    M = np.load(os.path.join(golden_dir, "masks.npz"))
    pred, flip = _mask_inputs()
    for nf in (3, 5):
        for sig in (False, True):
            _check_mask(M, "var%d%s" % (nf, "s" if sig else ""), losses.var_mask(pred, flip, nf, sig))
    _check_mask(M, "grad", losses.grad_mask(pred))
    _check_mask(M, "grad_thr", losses.grad_mask(pred, 0.2, 0.85))


def test_grad_stencil_matches_numpy():
    x = np.random.default_rng(0).normal(size=(8, 5, 5)).astype(np.float32)
    assert np.array_equal(losses._grad_t(x), np.gradient(x, axis=0))


def test_state_keys():
    keys = spec.state_dict_keys(24)
    assert len(keys) == 293
    n = sum(int(np.prod(s)) for s in spec.param_shapes(24).values())
    assert n == 48003705          # 48 003 705 trainable parameters
    assert len(spec.param_shapes(24)) == 158


_SPENT = [0.0]          # seconds of oracle time the full-step fixtures have taken so far in this process
STEP_CASES = ["step_bv5", "step_gv_pseudo", "step_bvgv3", "step_jhmdb_bv", "step_bv5_bs8", "step_gv_bs8", "step_jhmdb_bv_bs8", "step_gv_pseudo_bs8", "step_refinit_bv5"]


@pytest.mark.parametrize("tag", STEP_CASES)
def test_full_step(golden_dir, tag):
    path = os.path.join(golden_dir, tag + ".npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    # every fixture runs by default inside a time budget (PICONS_SLOW_BUDGET_S, default 240 s of oracle time over the whole parametrised
    # test: ~13 s per bs = 2 case and ~45 s per bs = 8 case on 8 threads); what does not fit is skipped and says so.  PICONS_SLOW=1: no budget.
    budget = float(os.environ.get("PICONS_SLOW_BUDGET_S", "240"))
    if tag != "step_bv5" and not os.environ.get("PICONS_SLOW") and _SPENT[0] > budget:
        pytest.skip("oracle time budget of %.0f s used up (%.0f s): set PICONS_SLOW=1 to run every fixture" % (budget, _SPENT[0]))
    import time as _time
    _t0 = _time.time()
    S = np.load(path)
    ncls = int(S["num_classes"]); epoch = int(S["epoch"]); stepid = int(S["stepid"])
    akw = dict(ast.literal_eval(str(S["args"])))
    args = ostep.default_args(**akw)
    bs = int(S["bs"]) if "bs" in S.files else 2
    conditioned = bool(int(S["conditioned"])) if "conditioned" in S.files else True
    state = synthetic.init_state(seed=47, num_classes=ncls, conditioned=conditioned)
    P = ostep.as_torch_params(state)
    lab, unl, perm, drops = synthetic.make_step_inputs(bs, rank=0, step=stepid, num_classes=ncls)
    torch.set_num_threads(int(S["_threads"]))
    r = ostep.train_step(P, args, lab, unl, epoch, float(S["ramp"]), perm, drops)
    r["total"].backward()
    _SPENT[0] += _time.time() - _t0
    # bars from BASELINE.json north_star: logits / masks 1e-3, loss scalars 1e-4.  With the reference's own init (SURVEY
    # finding 4) fp32 runs of the same code differ by more than that between thread counts; that fixture was produced at the
    # thread count recorded in it, and gets the bars the reference meets against its own fp64 run (kept in the fixture)
    t_out, t_loss = (1e-3, 1e-4) if conditioned else (5e-3, 5e-3)
    close(r["predicted_action"], S["predicted_action"], t_out, what="logits")
    close(r["output"][:, :, :, ::8, ::8], S["output_sample"], t_out, what="mask logits")
    for k in ("total", "loc", "cls", "cons"):
        close(r[k], S[k], t_loss, what=k)
    gn = dict(zip([str(x) for x in S["grad_names"]], S["grad_norms"]))
    for n, ref in gn.items():
        got = float(P[n].grad.norm())
        assert abs(got - ref) <= 2e-2 * max(ref, 1e-6) + 1e-7, (n, got, ref)
    for k in S.files:
        if k.startswith("grad::"):
            ref = S[k]
            close(P[k[6:]].grad, ref, 2e-2 * np.abs(ref).max() + 1e-8, what=k)
        if k.startswith("gsample::"):
            ref = S[k]
            stride = {"upsample3.weight": 7, "upsample4.weight": 13, "primary_caps.pose.weight": 997, "primary_caps.a.weight": 97}[k[9:]]
            close(P[k[9:]].grad.reshape(-1)[::stride], ref, 2e-2 * np.abs(ref).max() + 1e-8, what=k)
        if k.startswith("buf::"):
            close(P[k[5:]], S[k], 1e-5, what=k)


def test_training_trajectory(golden_dir):
    """Three steps of the reference's own training loop (main_ucf101.py:171-184 with optim.Adam(lr 1e-4, eps 1e-6) of :416, a fresh
    minibatch per step; tests/golden/traj_bv5.npz from tools/make_goldens.py --only traj) against the oracle's step + its Adam restatement:
    Adam moments and bias correction at t > 1, BatchNorm running statistics after six forward passes, num_batches_tracked."""
    path = os.path.join(golden_dir, "traj_bv5.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    S = np.load(path)
    bs, nsteps, lr = int(S["bs"]), int(S["nsteps"]), float(S["lr"])
    args = ostep.default_args(**dict(ast.literal_eval(str(S["args"]))))
    state = synthetic.init_state(seed=47, num_classes=24, conditioned=True)
    P = ostep.as_torch_params(state)
    torch.set_num_threads(int(S["_threads"]))
    m, v = {}, {}
    for s_ in range(nsteps):
        lab, unl, perm, drops = synthetic.make_step_inputs(bs, rank=0, step=int(S["stepids"][s_]), num_classes=24)
        for p in P.values():
            p.grad = None
        r = ostep.train_step(P, args, lab, unl, int(S["epoch"]), float(S["ramp"]), perm, drops)
        r["total"].backward()
        ostep.adam_step(P, m, v, s_ + 1, lr)
        # measured at generation time: step 0 identical, step 1 7.6e-6, step 2 5e-5 on the total (the class loss carries it: Adam's first update
        # is +-lr per element whatever the gradient's size, and the few elements whose gradient is rounding noise around zero move the other
        # way here than in the reference -- 0.1 % of Mixed_4f.b0's weights end 2 lr apart)
        for k in ("total", "loc", "cls", "cons"):
            close(r[k], S["s%d::%s" % (s_, k)], 1e-4, what="step %d %s" % (s_, k))
        close(r["predicted_action"], S["s%d::predicted_action" % s_], 1e-3, what="step %d logits" % s_)
    for k in S.files:
        if k.startswith("buf::") and not k.endswith("num_batches_tracked"):
            close(P[k[5:]], S[k], 2e-5, what=k)
        if k.startswith("param::"):
            ref = S[k]
            init = np.asarray(state[k[7:]])
            # the parameter moved by about nsteps * lr per element; sign-noise elements (above) are off by up to 2 * lr each
            assert np.abs(P[k[7:]].detach().numpy() - ref).max() <= 2.5 * lr * nsteps, k
            moved = np.abs(ref - init).mean()
            assert np.abs(P[k[7:]].detach().numpy() - ref).mean() <= 0.05 * moved + 1e-7, (k, moved)
