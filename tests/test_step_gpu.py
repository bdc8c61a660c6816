"""GPU parity of the whole fused train step (plan replayed through libpicons.so):
 - against the CPU oracle on identical synthetic clips / weights / shuffle / dropout draws at a
   reduced frame size the oracle finishes in seconds, every parameter gradient included;
 - against THE REFERENCE'S OWN outputs (tests/golden/step_*.npz) at the full 8x224x224 size.
Bars (BASELINE.json north_star): logits and localisation masks 1e-3, loss scalars 1e-4 (fp32)."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import step as ostep
from picons_amd import capi, spec, step as pstep, synthetic

pytestmark = pytest.mark.gpu


def run_pair(akw, hw, bs, epoch, ncls=24, jhmdb=False, stepid=0, lr=1e-4, conditioned=True):
    args = pstep.default_args(lr=lr, **akw)
    state = synthetic.init_state(47, ncls, conditioned=conditioned)
    eng = pstep.StepEngine(args, bs=bs, hw=hw, num_classes=ncls, jhmdb=jhmdb, state=state)
    lab, unl, perm, drops = synthetic.make_step_inputs(bs, rank=0, step=stepid, num_classes=ncls, hw=hw)
    ramp = pstep.exp_rampup(100)(epoch)
    eng.stage(lab, unl, perm, drops)
    eng.forward_backward(epoch, ramp)
    torch.cuda.synchronize()
    P = ostep.as_torch_params(state)
    oa = ostep.default_args(dataset="jhmdb" if jhmdb else "ucf101", **akw)
    ref = ostep.train_step(P, oa, lab, unl, epoch, ramp, perm, drops)
    ref["total"].backward()
    # fp64 run of the same oracle: the anchor for gradients (the fp32 reference differs from ITSELF by ~1 % on
    # gradients across thread counts, SURVEY finding 4, so fp32-vs-fp32 bars tighter than that are meaningless)
    P64 = ostep.as_torch_params(state, dtype=torch.float64)
    ref64 = ostep.train_step(P64, oa, lab, unl, epoch, ramp, perm, drops, dtype=torch.float64)
    ref64["total"].backward()
    return eng, ref, P, P64


def check_gradients_fp64_anchored(eng, P, P64, tag, floor=5e-3):
    """Every parameter gradient: relative L2 per tensor against the fp64 oracle, judged next to the fp32 oracle's own
    distance from it (the fp32 reference differs from ITSELF by ~1 % on gradients, SURVEY finding 4)."""
    bad, rows = [], []
    num_g = num_c = den_all = 0.0
    for name in eng.plan.pshape:
        g = eng.grad(name).cpu().double(); r32 = P[name].grad.double(); r64 = P64[name].grad
        den = r64.norm().item() + 1e-12
        rel_g = (g - r64).norm().item() / den
        rel_c = (r32 - r64).norm().item() / den
        rows.append((name, rel_g, rel_c, den))
        num_g += (g - r64).norm().item() ** 2; num_c += (r32 - r64).norm().item() ** 2; den_all += den ** 2
        if rel_g > max(4 * rel_c, floor) and (g - r64).abs().max().item() > 1e-7:
            bad.append((name, rel_g, rel_c, den))
    tot_g, tot_c = (num_g / den_all) ** 0.5, (num_c / den_all) ** 0.5
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/step_grad_err_%s.txt" % tag, "w") as f:
        f.write("whole-gradient rel-L2 vs fp64 oracle: hip %.3e   fp32-oracle %.3e\n" % (tot_g, tot_c))
        for r in sorted(rows, key=lambda r: -r[1])[:40]:
            f.write("%-44s hip %.3e  cpu32 %.3e  |g| %.3e\n" % r)
    assert not bad, bad[:10]
    assert tot_g <= max(3 * tot_c, 2e-3), (tot_g, tot_c)
    return rows


CASES = [
    ("bv5", dict(bv=True, n_frames=5, wt_cons=0.1), 1, 24, False),
    ("gv_pseudo", dict(gv=True, lower_thresh=0.2, upper_thresh=0.9, wt_cons=0.1), 12, 24, False),
    ("bvgv3_sig", dict(bv=True, gv=True, n_frames=3, predict_maps=True), 3, 24, False),
    ("jhmdb_bv", dict(bv=True, n_frames=5, wt_cons=0.1), 1, 21, True),
    ("l2_only", dict(), 1, 24, False),
]


@pytest.mark.parametrize("tag,akw,epoch,ncls,jhmdb", CASES)
def test_step_vs_oracle_small(tag, akw, epoch, ncls, jhmdb):
    hw, bs = 112, 2
    eng, ref, P, P64 = run_pair(akw, hw, bs, epoch, ncls, jhmdb)
    got = eng.read_scalars()
    out, flip, pred = eng.outputs()
    for k in ("total", "loc", "cls", "cons"):
        assert abs(got[k] - float(ref[k])) <= 1e-4, (k, got[k], float(ref[k]))
    assert (pred.cpu() - ref["predicted_action"]).abs().max().item() <= 1e-3
    assert (out.cpu() - ref["output"]).abs().max().item() <= 1e-3
    assert (flip.cpu() - ref["flip_op"]).abs().max().item() <= 1e-3
    check_gradients_fp64_anchored(eng, P, P64, tag)
    # BN running statistics after the two passes
    for pre, _ci, co, _k, _s in spec.trunk_units()[:6] + spec.trunk_units()[-3:]:
        for nm in ("running_mean", "running_var"):
            key = pre + ".bn." + nm
            o = eng.plan.roff[key]
            assert (eng.R[o:o + co].cpu() - P[key]).abs().max().item() <= 1e-5, key
    # Adam update
    before = {k: eng.param(k).clone() for k in ("conv_caps.beta_u", "smooth.weight", "conv1.Mixed_4f.b0.bn.weight")}
    eng.adam(1e-4)
    m, v = {}, {}
    ostep.adam_step(P, m, v, 1, 1e-4)
    for k in before:
        assert (eng.param(k).cpu() - P[k].detach()).abs().max().item() <= 5e-5, k   # half an Adam step (lr 1e-4): |g| ~ eps entries amplify the ~2 % gradient noise


@pytest.mark.parametrize("bs,hw", [(6, 112), (4, 80)])
def test_step_vs_oracle_other_batch_and_frame_sizes(bs, hw):
    """Nothing in the path may assume a power-of-two batch or the two frame sizes the other tests use: bs = 6 (three labeled + three
    unlabeled clips, 12 clip-passes: ragged row tiles in every GEMM, 12 x 6 x 6 EM positions) at 112^2 and bs = 4 at 80^2
    (10 x 10 features, 2 x 2 capsule grid: the smallest frame size but one the 9 x 9 PrimaryCaps conv admits).  Scalars / outputs at the bars of test_step_vs_oracle_small; the per-tensor gradient floor is 2 % here: on
    this minibatch ONE pre-activation of Mixed_4f.b2b (channel 0) sits within fp32 rounding of zero, its ReLU mask comes out the other
    way than in the fp64 run, and that single element moves the branch's small gradients by 0.6 - 1 % (every other channel agrees to
    1e-6: tools/probe_bs_grads.py 6 112 7 conv1.Mixed_4f.b2b.bn.bias) -- a discontinuity of the function, not of the kernels."""
    akw = dict(bv=True, gv=True, n_frames=5, wt_cons=0.1)
    eng, ref, P, P64 = run_pair(akw, hw, bs, 1, 24, False, stepid=7)
    got = eng.read_scalars()
    out, flip, pred = eng.outputs()
    for k in ("total", "loc", "cls", "cons"):
        assert abs(got[k] - float(ref[k])) <= 1e-4, (k, got[k], float(ref[k]))
    assert (pred.cpu() - ref["predicted_action"]).abs().max().item() <= 1e-3
    assert (out.cpu() - ref["output"]).abs().max().item() <= 1e-3
    assert (flip.cpu() - ref["flip_op"]).abs().max().item() <= 1e-3
    check_gradients_fp64_anchored(eng, P, P64, "bs%d_hw%d" % (bs, hw), floor=2e-2)


GSAMPLE_STRIDE = {"upsample3.weight": 7, "upsample4.weight": 13, "primary_caps.pose.weight": 997, "primary_caps.a.weight": 97}


@pytest.mark.parametrize("tag", ["step_bv5", "step_gv_pseudo", "step_bvgv3", "step_jhmdb_bv", "step_bv5_bs8", "step_gv_bs8",
                                 "step_jhmdb_bv_bs8", "step_gv_pseudo_bs8"])
def test_step_vs_reference_golden_full_size(golden_dir, tag):
    """HIP step against THE REFERENCE'S OWN outputs.  The two *_bs8 fixtures are BASELINE configs[1] (--bv --n_frames 5)
    and configs[2] (--gv, thresholds None) at the batch size the metric is quoted on: 4 labeled + 4 unlabeled clips, i.e. the
    B-dependent semantics (gv (B,B,...) broadcast utils/losses.py:74-76, joint Dice :44-57, Spread /b^2 :34-35, BN statistics
    over 8 clips per pass pytorch_i3d.py:116-119); step_jhmdb_bv_bs8 is BASELINE configs[4]'s per-rank workload (21 classes,
    main_jhmdb.py:50-140) and step_gv_pseudo_bs8 the epoch >= thresh_epoch branch (argmax pseudo-labels, capsules_ucf101.py:463),
    the two other workloads bench.py times at bs = 8."""
    S = np.load(os.path.join(golden_dir, tag + ".npz"))
    ncls = int(S["num_classes"]); epoch = int(S["epoch"]); stepid = int(S["stepid"])
    bs = int(S["bs"]) if "bs" in S.files else 2
    akw = dict(ast.literal_eval(str(S["args"])))
    jh = akw.pop("dataset", "ucf101") == "jhmdb"
    for k in ("wt_seg",):
        akw.pop(k, None)
    args = pstep.default_args(**akw)
    eng = pstep.StepEngine(args, bs=bs, hw=224, num_classes=ncls, jhmdb=jh)
    lab, unl, perm, drops = synthetic.make_step_inputs(bs, rank=0, step=stepid, num_classes=ncls)
    eng.stage(lab, unl, perm, drops)
    eng.forward_backward(epoch, float(S["ramp"]))
    got = eng.read_scalars()
    out, _flip, pred = eng.outputs()
    for k in ("total", "loc", "cls", "cons"):
        assert abs(got[k] - float(S[k])) <= 1e-4, (k, got[k], float(S[k]))
    assert np.abs(pred.cpu().numpy() - S["predicted_action"]).max() <= 1e-3
    assert np.abs(out[:, :, :, ::8, ::8].cpu().numpy() - S["output_sample"]).max() <= 1e-3
    assert np.abs(out.sum(dim=(-1, -2)).cpu().numpy() - S["output_frame_sum"]).max() <= 1e-3 * 224 * 224
    gn = dict(zip([str(x) for x in S["grad_names"]], S["grad_norms"]))
    # gradient norms within 2 % of the fp32 reference's.  The PrimaryCaps activation-capsule parameters are the
    # worst-conditioned in the model (their gradient passes through all of EM routing's sigmoid / log terms): the fp32
    # reference itself is 4.6-5.1 % (rel-L2) from an fp64 run there and this path 4.2-4.4 % (test_step_vs_oracle_small,
    # gpurun_out/step_grad_err_*.txt); at this size fp64 gives |g| = 1.7307e-5, the reference 1.7111..1.7225e-5 depending
    # on its thread count, so they get the noise-sized bar
    loose = {"primary_caps.a.weight": 6e-2, "primary_caps.a.bias": 6e-2}
    bad = [(n, float(eng.grad(n).norm()), r) for n, r in gn.items()
           if abs(float(eng.grad(n).norm()) - r) > loose.get(n, 2e-2) * max(r, 1e-6) + 1e-7]
    assert not bad, bad[:10]
    # element-wise against the fp32 reference's own gradients: 2 % of the tensor's max at bs = 2; 5 % at bs = 8, where both this
    # path and the fp32 reference are 1.2-1.4 % (whole-gradient rel-L2) from an fp64 run (gpurun_out/step_grad_err_*_bs8.txt) --
    # the tight per-tensor bar at that size is the fp64-anchored one of test_step_bs8_full_size_vs_oracle
    #
    # Round 5: the bs = 2 fixtures also hold an fp64 run of the reference (tools/make_goldens.py --only steps_f64; their fp32 contents are
    # unchanged).  On these tensors the reference's own fp32 run is 0.6 - 1.8 % (max-relative) from it -- the size of the 2 % bar above -- so a
    # second, anchored bar stands beside it: no further from the fp64 run than 3x the fp32 reference's own distance (floor 1 %), the rule of
    # test_training_trajectory_vs_reference (measured: at most 1.95x, `Conv3d_1a_7x7.bn.weight` of step_gv_pseudo).  Both bars caught the one launch that may not run in Winograd F(4x4, 3x3): with Conv3d_2c's
    # FORWARD in it the stem's BatchNorm gradients went from 1.0 - 1.2x to 2.0 - 2.5x the fp32 reference's distance and over the 2 % bar (EM routing amplifies any perturbation of the
    # trunk's forward); with that launch in F(2x2, 3x3) and the other five in F(4x4, 3x3) every number below is what it was before
    # (tools/probe_grad_margin.py, profiles/r05_wino4_grad_margin.txt).
    gtol = 2e-2 if bs == 2 else 5e-2
    anchored = any(k.startswith("f64::grad::") for k in S.files)
    for k in S.files:
        if k.startswith("grad::"):
            ref, g = S[k], eng.grad(k[6:]).cpu().numpy()
            assert np.abs(g - ref).max() <= gtol * np.abs(ref).max() + 1e-8, k
            if anchored:
                g64 = S["f64::" + k]
                e_hip, e_ref = np.abs(g - g64).max() / np.abs(g64).max(), np.abs(ref - g64).max() / np.abs(g64).max()
                assert e_hip <= max(3 * e_ref, 1e-2) + 1e-8 / np.abs(g64).max(), (k, e_hip, e_ref)
        if k.startswith("gsample::"):        # strided samples of the big 224-only / PrimaryCaps gradients
            ref = S[k]
            g = eng.grad(k[9:]).reshape(-1)[::GSAMPLE_STRIDE[k[9:]]].cpu().numpy()
            assert np.abs(g - ref).max() <= gtol * np.abs(ref).max() + 1e-8, k
        if k.startswith("buf::") and not k.endswith("num_batches_tracked"):
            o = eng.plan.roff[k[5:]]
            assert np.abs(eng.R[o:o + S[k].size].cpu().numpy() - S[k]).max() <= 1e-5, k


BS8 = [("bv5_bs8", dict(bv=True, n_frames=5, wt_cons=0.1), 4, 1, 24, False), ("gv_bs8", dict(gv=True, wt_cons=0.1), 5, 1, 24, False),
       # configs[4] per rank (21-class JHMDB step) and the pseudo-label branch (epoch 12 >= thresh_epoch 11), as bench.py --jhmdb /
       # --epoch 12 time them
       ("jhmdb_bv_bs8", dict(bv=True, n_frames=5, wt_cons=0.1), 7, 1, 21, True),
       ("gv_pseudo_bs8", dict(gv=True, lower_thresh=0.2, upper_thresh=0.9, wt_cons=0.1), 8, 12, 24, False)]


@pytest.mark.parametrize("tag,akw,stepid,epoch,ncls,jhmdb", BS8)
def test_step_bs8_full_size_vs_oracle(tag, akw, stepid, epoch, ncls, jhmdb):
    """BASELINE configs[1] / configs[2] exactly as bench.py runs them (bs = 8, 8x224x224, epoch 1) against the CPU oracle on
    the same clips: loss scalars 1e-4, logits / masks 1e-3, and EVERY parameter gradient per tensor against an fp64 run of
    the oracle -- this is where the 224-only kernels (the stem's wgrad4, conv112's row-segment wgrad, the 256-column
    wgrad tiles, upsample3 / the merged tail, the spectral PrimaryCaps at M = 6400) meet a per-tensor bar."""
    eng, ref, P, P64 = run_pair(akw, 224, 8, epoch, ncls, jhmdb, stepid=stepid)
    got = eng.read_scalars()
    out, flip, pred = eng.outputs()
    for k in ("total", "loc", "cls", "cons"):
        assert abs(got[k] - float(ref[k])) <= 1e-4, (k, got[k], float(ref[k]))
    assert (pred.cpu() - ref["predicted_action"]).abs().max().item() <= 1e-3
    assert (out.cpu() - ref["output"]).abs().max().item() <= 1e-3
    assert (flip.cpu() - ref["flip_op"]).abs().max().item() <= 1e-3
    rows = {r[0]: r for r in check_gradients_fp64_anchored(eng, P, P64, tag)}
    for name in ("conv1.Conv3d_1a_7x7.conv3d.weight", "conv112.weight", "upsample3.weight", "upsample4.weight",
                 "smooth.weight", "primary_caps.pose.weight", "primary_caps.a.weight"):
        assert name in rows
    for pre, _ci, co, _k, _s in spec.trunk_units()[:4] + spec.trunk_units()[-3:]:       # BN statistics over 8 clips per pass
        for nm in ("running_mean", "running_var"):
            key = pre + ".bn." + nm
            o = eng.plan.roff[key]
            assert (eng.R[o:o + co].cpu() - P[key]).abs().max().item() <= 1e-5, key


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-300))


def test_step_reference_init_vs_reference_fp64(golden_dir):
    """SURVEY 8(c) "reference-init case": the reference's OWN initialisation (PrimaryCaps weights N(0, 0.1), ConvCaps.weights
    randn: capsules_ucf101.py:36,39,103 -- what real checkpoints look like), where EM routing amplifies fp32 rounding so
    far that the fp32 reference disagrees with itself at the 1e-3 level (SURVEY finding 4).  The fixture holds an fp32 AND an
    fp64 run of the reference itself; the HIP step is judged in rel-L2 against the fp64 run: <= 1e-3, and no further from
    it than 2x the fp32 reference's own distance."""
    S = np.load(os.path.join(golden_dir, "step_refinit_bv5.npz"))
    assert int(S["conditioned"]) == 0
    bs, stepid, epoch = int(S["bs"]), int(S["stepid"]), int(S["epoch"])
    akw = dict(ast.literal_eval(str(S["args"])))
    akw.pop("dataset", None)
    eng = pstep.StepEngine(pstep.default_args(**akw), bs=bs, hw=224, state=synthetic.init_state(47, 24, conditioned=False))
    lab, unl, perm, drops = synthetic.make_step_inputs(bs, rank=0, step=stepid)
    eng.stage(lab, unl, perm, drops)
    eng.forward_backward(epoch, float(S["ramp"]))
    got = eng.read_scalars()
    out, _flip, pred = eng.outputs()
    lines = []
    for key, mine in (("predicted_action", pred.cpu().numpy()), ("output_sample", out[:, :, :, ::8, ::8].cpu().numpy()),
                      ("output_frame_sum", out.sum(dim=(-1, -2)).cpu().numpy())):
        r_hip, r_ref = _rel(mine, S["f64::" + key]), _rel(S[key], S["f64::" + key])
        lines.append("%-20s rel-L2 vs fp64 reference: hip %.3e   fp32 reference %.3e" % (key, r_hip, r_ref))
        assert r_hip <= 1e-3, (key, r_hip, r_ref)
        assert r_hip <= max(2 * r_ref, 2e-5), (key, r_hip, r_ref)
    for k in ("total", "loc", "cls", "cons"):
        d_hip, d_ref = abs(got[k] - float(S["f64::" + k])), abs(float(S[k]) - float(S["f64::" + k]))
        lines.append("%-20s |d| vs fp64 reference: hip %.3e   fp32 reference %.3e   value %.6f" % (k, d_hip, d_ref, float(S["f64::" + k])))
        assert d_hip <= max(1e-3 * abs(float(S["f64::" + k])), 2 * d_ref, 1e-5), (k, d_hip, d_ref)
    names = [str(x) for x in S["grad_names"]]
    n32, n64 = dict(zip(names, S["grad_norms"])), dict(zip(names, S["f64::grad_norms"]))
    worst, fails = [], []
    for n in names:
        d_hip, d_ref = abs(float(eng.grad(n).norm()) - n64[n]), abs(n32[n] - n64[n])
        worst.append((d_hip / max(n64[n], 1e-30), d_ref / max(n64[n], 1e-30), n))
        if not d_hip <= max(2 * d_ref, 2e-2 * n64[n]) + 1e-9:
            fails.append(("norm", n, d_hip, d_ref, n64[n]))
    for k in S.files:
        if k.startswith("f64::grad::") or k.startswith("f64::gsample::"):
            name = k.split("::")[2]
            g = eng.grad(name).reshape(-1)
            if "gsample" in k:
                g = g[::GSAMPLE_STRIDE[name]]
            r_hip, r_ref = _rel(g.cpu().numpy(), S[k].reshape(-1)), _rel(S[k[5:]].reshape(-1), S[k].reshape(-1))
            lines.append("%-44s grad rel-L2 vs fp64 reference: hip %.3e   fp32 reference %.3e" % (name, r_hip, r_ref))
            if not r_hip <= max(2 * r_ref, 5e-3):
                fails.append(("tensor", name, r_hip, r_ref))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/step_refinit.txt", "w") as f:
        f.write("\n".join(lines) + "\n")
        for w in sorted(worst, reverse=True)[:40]:
            f.write("|g| rel. error vs fp64: hip %.3e  fp32 reference %.3e  %s\n" % w)
    assert not fails, fails[:10]


def test_smoke_entry():
    import __graft_entry__ as entry
    entry.smoke_check(hw=112)


def test_segmented_backward_matches_unsegmented():
    """The data-parallel path replays the backward list in bucket segments (StepEngine.forward_backward with a
    reducer).  With a stand-in reducer that only records launches, gradients must be bit-identical to the
    single-call replay for everything that does not use float atomics, and the buckets must be launched in order."""
    args = pstep.default_args(bv=True, n_frames=5, wt_cons=0.1)
    eng = pstep.StepEngine(args, bs=2, hw=112)
    lab, unl, perm, drops = synthetic.make_step_inputs(2, hw=112)
    eng.stage(lab, unl, perm, drops)
    eng.forward_backward(1, 0.01)
    torch.cuda.synchronize()
    g0 = eng.G.clone()

    class FakeReducer:
        world = 2
        active = True
        gscale = 0.5

        def __init__(self, buckets):
            self.buckets = buckets
            self.launched = []

        def launch(self, i, streams=()):
            self.launched.append(i)

        def wait(self):
            pass
    fr = FakeReducer(eng.plan.grad_buckets(3_000_000))
    R0 = eng.R.clone()
    eng.load_state(synthetic.init_state(47, 24))          # running stats back to their initial values
    eng.forward_backward(1, 0.01, reducer=fr)
    torch.cuda.synchronize()
    assert fr.launched == list(range(len(fr.buckets)))
    rel = ((eng.G - g0).norm() / g0.norm()).item()
    assert rel < 1e-4, rel                                  # wgrad split-K atomics reorder fp32 sums only
    bn = eng.plan.poff["conv1.Mixed_4f.b0.bn.weight"]
    assert torch.equal(eng.G[bn:bn + 256], g0[bn:bn + 256])  # BN / bias gradients use no atomics: bit-identical
    assert torch.allclose(eng.R, R0)


def test_multi_lane_replay_matches_single_lane():
    """The default engine replays the Inception branches on four HIP streams (plan lanes, FORK/JOIN ops).
    Same inputs through a single-lane engine: outputs, loss scalars, running statistics and gradients
    equal up to the order of fp32 atomic sums (the tail's column sums forward, split-K wgrad backward),
    i.e. well inside the parity bars -- on three different minibatches, so a lane race
    that only shows up under a particular timing has three chances."""
    args = pstep.default_args(bv=True, gv=True, n_frames=3, wt_cons=0.1)
    e1 = pstep.StepEngine(args, bs=2, hw=112, lanes=1)
    e4 = pstep.StepEngine(args, bs=2, hw=112, lanes=4)   # the default is 2; 4 exercises every lane
    assert len(e4.side) == 3 and not e1.side
    for stepid in range(3):
        lab, unl, perm, drops = synthetic.make_step_inputs(2, step=stepid, hw=112)
        res = []
        for eng in (e1, e4):
            eng.stage(lab, unl, perm, drops)
            eng.forward_backward(1, 0.01)
            torch.cuda.synchronize()
            res.append((eng.read_scalars(), [t.clone() for t in eng.outputs()], eng.G.clone(), eng.R.clone()))
        (s1, o1, g1, r1), (s4, o4, g4, r4) = res
        for a, b in zip(o1, o4):
            assert (a - b).abs().max().item() < 5e-5
        for k in s1:
            assert abs(s1[k] - s4[k]) <= 1e-5 * max(1.0, abs(s1[k])), (k, s1[k], s4[k])
        assert torch.allclose(r1, r4, rtol=1e-6, atol=1e-7)
        rel = ((g1 - g4).norm() / g1.norm()).item()
        assert rel < 2e-4, (stepid, rel)
    # (no Adam between the steps: its first update is sign-like, so fp32-noise-sized gradient differences
    # would turn into lr-sized parameter differences and the engines would legitimately drift apart)


def test_stacked_1x1_units_match_separate_units(monkeypatch):
    """The three 1x1x1 Unit3Ds of every Inception module that read the module input run as ONE conv + BN + ReLU over
    their stacked channels, in a buffer wider than the module output (plan.inception).  Against an engine that keeps
    them separate (PICONS_FUSE1X1=0): same outputs, scalars, running statistics and per-parameter gradients -- the flat
    buffers are laid out differently, so everything is compared by name."""
    args = pstep.default_args(bv=True, gv=True, n_frames=3, wt_cons=0.1)
    monkeypatch.setenv("PICONS_FUSE1X1", "0")
    e0 = pstep.StepEngine(args, bs=2, hw=112)
    monkeypatch.setenv("PICONS_FUSE1X1", "1")
    e1 = pstep.StepEngine(args, bs=2, hw=112)
    assert not e0.plan.fused_groups and len(e1.plan.fused_groups) == 7
    assert e0.plan.poff != e1.plan.poff and e0.plan.nparams == e1.plan.nparams
    e1.load_state({k: v.cpu() for k, v in e0.state_dict().items()})
    lab, unl, perm, drops = synthetic.make_step_inputs(2, step=1, hw=112)
    res = []
    for eng in (e0, e1):
        eng.stage(lab, unl, perm, drops)
        eng.forward_backward(1, 0.01)
        torch.cuda.synchronize()
        res.append((eng.read_scalars(), [t.clone() for t in eng.outputs()]))
    (s0, o0), (s1, o1) = res
    for a, b in zip(o0, o1):
        assert (a - b).abs().max().item() < 5e-5
    for k in s0:
        assert abs(s0[k] - s1[k]) <= 1e-5 * max(1.0, abs(s0[k])), (k, s0[k], s1[k])
    sd0, sd1 = e0.state_dict(), e1.state_dict()
    for k in sd0:
        if "running_" in k:
            assert torch.allclose(sd0[k], sd1[k], rtol=1e-6, atol=1e-7), k
    num = den = 0.0
    for k in e0.plan.pshape:
        g0, g1 = e0.grad(k), e1.grad(k)
        num += float((g0 - g1).double().pow(2).sum()); den += float(g0.double().pow(2).sum())
        if ".bn." in k:          # BN gradients: sums over the same rows, reduced in the same order
            assert torch.allclose(g0, g1, rtol=2e-4, atol=1e-6), k
    assert (num / den) ** 0.5 < 2e-4


def test_batchnorm_finalize_folded_into_apply_matches_default():
    """PICONS_BN_FUSED (round 5; an EXPERIMENT switch -- no gain measured: passed explicitly, the environment alone does not select it,
    picons_amd/switches.py): the 28 x 28 layers' BatchNorm statistics are taken in the prologue of the
    apply kernels, forward (PC_OP_BN_FIN_APPLY in the plan) and backward (pc_bn_bwd's two-launch form).  Against the default engine on the same
    minibatch: same outputs, scalars, running statistics and gradients (the per-channel sums are taken in fp64 in another fixed order)."""
    args = pstep.default_args(bv=True, n_frames=5, wt_cons=0.1)
    lab, unl, perm, drops = synthetic.make_step_inputs(2, step=2, hw=112)
    res = []
    for fused in ("0", "1"):
        eng = pstep.StepEngine(args, bs=2, hw=112, exp={"PICONS_BN_FUSED": fused})
        nfa = sum(1 for op in eng.plan.lists["fwd"] if op[0] == capi.OP_BN_FIN_APPLY)
        assert (nfa > 10) if fused == "1" else (nfa == 0)
        eng.stage(lab, unl, perm, drops)
        eng.forward_backward(1, 0.01)
        torch.cuda.synchronize()
        res.append((eng.read_scalars(), [t.clone() for t in eng.outputs()], eng.state_dict(), eng.G.clone()))
    (s0, o0, sd0, g0), (s1, o1, sd1, g1) = res
    for a, b in zip(o0, o1):
        assert (a - b).abs().max().item() < 5e-5
    for k in s0:
        assert abs(s0[k] - s1[k]) <= 1e-5 * max(1.0, abs(s0[k])), (k, s0[k], s1[k])
    for k in sd0:
        if "running_" in k:
            assert torch.allclose(sd0[k], sd1[k], rtol=1e-6, atol=1e-7), k
    assert ((g0 - g1).norm() / g0.norm()).item() < 2e-4


def test_step_is_deterministic_run_to_run():
    """SURVEY 5 determinism check (catches races in the reductions): the same step twice from the same state.  No forward
    kernel uses float atomics, so outputs, logits, loss scalars and BN running statistics must be BIT-identical, and so must
    every gradient that is reduced without atomics (BatchNorm, biases of conv / transposed-conv layers, all of ConvCaps:
    em_bwd's partials have one owner thread per element and a fixed-order final sum) -- since round 6 that includes every split-K
    weight gradient (K-slice images added in slice order) and the merged tail (per-class K-slice images, per-block bias rows, a block-owned
    smooth-weight reduction): no gradient of the step adds fp32 partials in arrival order any more, the whole flat gradient is compared
    with torch.equal."""
    args = pstep.default_args(bv=True, gv=True, n_frames=5, wt_cons=0.1)
    eng = pstep.StepEngine(args, bs=2, hw=112)
    lab, unl, perm, drops = synthetic.make_step_inputs(2, step=2, hw=112)
    runs = []
    for _ in range(3):
        eng.load_state(synthetic.init_state(47, 24))
        eng.stage(lab, unl, perm, drops)
        eng.forward_backward(1, 0.01)
        torch.cuda.synchronize()
        runs.append(([t.clone() for t in eng.outputs()], eng.aview(eng.plan.scalars, 20).clone(), eng.R.clone(), eng.G.clone()))
    for other in runs[1:]:
        for a, b in zip(runs[0][0], other[0]):
            assert torch.equal(a, b)
        assert torch.equal(runs[0][1], other[1]) and torch.equal(runs[0][2], other[2])
        for name in eng.plan.pshape:
            o = eng.plan.poff[name]; n = int(np.prod(eng.plan.pshape[name]))
            assert torch.equal(runs[0][3][o:o + n], other[3][o:o + n]), name


def test_step_is_deterministic_at_the_benchmark_size():
    """The same check on bench.py's workload (bs = 8, 8 x 224 x 224, both consistency losses): the K-slice counts, the folds of launches with more than
    128 slices and the merged tail's class sums are those of the measured step, not of the 112 x 112 case above.  Every output, scalar, running
    statistic and gradient of three runs from the same state is compared with torch.equal."""
    args = pstep.default_args(bv=True, gv=True, n_frames=5, wt_cons=0.1)
    eng = pstep.StepEngine(args, bs=8, hw=224)
    lab, unl, perm, drops = synthetic.make_step_inputs(8, step=3, hw=224)
    first = None
    for it in range(3):
        eng.load_state(synthetic.init_state(47, 24))
        eng.stage(lab, unl, perm, drops)
        eng.forward_backward(1, 0.01)
        torch.cuda.synchronize()
        got = [t.clone() for t in eng.outputs()] + [eng.aview(eng.plan.scalars, 20).clone(), eng.R.clone(), eng.G.clone()]
        if first is None:
            first = got
            assert torch.isfinite(first[-1]).all() and float(first[-1].abs().max()) > 0
            continue
        for q, (a, b) in enumerate(zip(first, got)):
            if not torch.equal(a, b):
                bad = [nm for nm in eng.plan.pshape
                       if not torch.equal(a[eng.plan.poff[nm]:eng.plan.poff[nm] + int(np.prod(eng.plan.pshape[nm]))],
                                          b[eng.plan.poff[nm]:eng.plan.poff[nm] + int(np.prod(eng.plan.pshape[nm]))])] if q == len(first) - 1 else []
                raise AssertionError("run %d differs from the first in item %d %s" % (it, q, bad[:6]))


def test_thread_events_can_be_released_between_steps():
    """pc_release_thread_events gives back the FORK / JOIN, fan-in and timing events of the calling thread; the next replay creates
    new ones and gives the same results."""
    from picons_amd import capi
    args = pstep.default_args(bv=True, n_frames=5, wt_cons=0.1)
    eng = pstep.StepEngine(args, bs=2, hw=112)
    eng.stage(*synthetic.make_step_inputs(2, step=5, hw=112))
    eng.forward_backward(1, 0.01, timed_kind=capi.OP_CONV)          # leaves timing pairs pending: dropped by the release
    a = eng.read_scalars()
    torch.cuda.synchronize()
    G0 = eng.G.clone()
    capi.call("pc_release_thread_events")
    eng.load_state(synthetic.init_state(47, 24))
    eng.forward_backward(1, 0.01)
    b = eng.read_scalars()
    torch.cuda.synchronize()
    assert a == b
    assert ((eng.G - G0).norm() / G0.norm()).item() < 2e-4
    eng.forward_backward(1, 0.01, timed_kind=capi.OP_CONV)
    eng.collect_timing()
    assert eng.kind_count > 50 and eng.kind_ms > 0.0
    capi.call("pc_release_thread_events")


def test_early_adam_uses_final_gradients():
    """The backward list's early Adam op (every parameter but the stem's, beside the stem's weight gradient) must see FINAL gradients:
    after one armed step from zero moments, every parameter equals p - lr * g / (|g| + eps) of the gradient left in G, and the same
    step with the op un-armed (PICONS_EARLY_ADAM=0 semantics: one Adam behind the backward) gives the same parameters."""
    args = pstep.default_args(bv=True, n_frames=5, wt_cons=0.1)
    eng = pstep.StepEngine(args, bs=2, hw=112)
    if eng.plan.op_adam_early is None:
        pytest.skip("no early Adam op in this configuration (one lane, or PICONS_EARLY_ADAM=0)")
    assert 0 < eng.plan.adam_split < eng.plan.nparams
    lab, unl, perm, drops = synthetic.make_step_inputs(2, step=3, hw=112)
    res = []
    for armed in (True, False):
        eng.load_state(synthetic.init_state(53, 24))
        eng.M.zero_(); eng.V.zero_(); eng.step_count = 0
        p0 = eng.P.clone()
        eng.stage(lab, unl, perm, drops)
        if armed:
            eng.run_staged(1, 0.01, lr=1e-3)
        else:
            eng.arm_early_adam(1e-3, on=False)
            eng.forward_backward(1, 0.01)
            eng.adam(1e-3)
        eng.synchronize()
        g = eng.G.double()
        want = p0.double() - 1e-3 * g / (g.abs() + 1e-6)
        err = (eng.P.double() - want).abs().max().item()
        assert err <= 2e-7, (armed, err)
        # the moments after the first step from zero, element by element: m = (1 - b1) g, v = (1 - b2) g^2 (Adam(betas 0.9, 0.999) of
        # main_ucf101.py:416) -- an element that either op visited twice, or not at all, is off by a factor here
        assert (eng.M.double() - 0.1 * g).abs().max().item() <= 1e-6 * max(1.0, g.abs().max().item()), armed
        assert ((eng.V.double() - 0.001 * g * g).abs() <= 1e-6 * (0.001 * g * g) + 1e-30).all(), armed
        assert int(eng.ops["bwd"][eng.plan.op_adam_early]["l"][0]) == 0          # disarmed again behind the step
        # ... and a SECOND step (t = 2: bias corrections 1 - 0.9^2, 1 - 0.999^2) on another minibatch
        l2, u2, pm2, dr2 = synthetic.make_step_inputs(2, step=4, hw=112)
        eng.stage(l2, u2, pm2, dr2)
        if armed:
            eng.run_staged(1, 0.01, lr=1e-3)
        else:
            eng.arm_early_adam(1e-3, on=False)
            eng.forward_backward(1, 0.01)
            eng.adam(1e-3)
        eng.synchronize()
        assert eng.step_count == 2
        res.append((eng.P.clone(), eng.M.clone(), eng.V.clone(), p0))
    # weight gradients are summed with fp32 atomics (arrival order), and EM routing amplifies the 1e-9 that leaves in the parameters after
    # step 1 into 1e-5 .. 1e-3 of the trunk's step-2 gradients (measured: rel-L2 1.8e-3 on the first moments, 5.9e-4 on the second; single
    # elements whose gradient is that noise move by up to 1.5 lr): the two schedules agree to that, not bit for bit -- both moments to 1e-2 in
    # rel-L2 and the parameters to 2 % of the distance they travelled in the two steps (the armed schedule splits the flat buffer between two
    # Adam ops: a range that one of them doubled, skipped or stepped with the wrong count shows here at 1e-1 .. 1)
    (Pa, Ma, Va, P0), (Pb, Mb, Vb, _) = res
    dP = ((Pa - Pb).norm() / (Pb - P0).norm()).item()
    dM, dV = ((Ma - Mb).norm() / Mb.norm()).item(), ((Va - Vb).norm() / Vb.norm()).item()
    print("early Adam vs one Adam after two steps: |dP| / |P - P0| %.3e, rel-L2 dM %.3e, dV %.3e" % (dP, dM, dV))
    assert dP < 2e-2 and dM < 1e-2 and dV < 1e-2, (dP, dM, dV)


@pytest.mark.parametrize("mode", ["default", "reducer"])
def test_training_trajectory_vs_reference(tmp_path, mode):
    """a17 beyond t = 1: THREE real steps -- a fresh minibatch each, Adam between them, four lanes -- against the reference's own
    loop run for three steps (tests/golden/traj_bv5.npz: main_ucf101.py:171-184 with optim.Adam(lr 1e-4, eps 1e-6) of :416).  `default`: the
    early-Adam op inside the backward list armed with a host-patched step count / lr every step; `reducer`: the DP schedule through a
    one-rank RCCL group, where that op covers the parameters whose buckets have been launched by then (behind those collectives, 1/world
    folded in) and the closing Adam the trunk's last bucket.  Checks the loss scalars of
    every step, the BatchNorm running statistics after six forward passes, num_batches_tracked and parameters after step 3."""
    import json
    import socket
    import subprocess
    import sys
    if os.environ.get("PICONS_LANES", "4") != "4":
        pytest.skip("the trajectory is pinned at the product default of four lanes (tools/gpu/switch_matrix.sh runs this file with PICONS_LANES=1)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    out = str(tmp_path / "traj.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "traj_worker.py"), mode, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900)
    assert p.returncode == 0 and os.path.exists(out), p.stdout[-3000:]
    v = json.load(open(out))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    json.dump(v, open(os.path.join(root, "gpurun_out", "trajectory_%s.json" % mode), "w"), indent=1)
    assert v["lanes"] == 4 and v["early_adam_op"] and v["step_count"] == 3
    # Step 1 starts from identical parameters: the loss scalars agree with the reference's fp32 run to 1e-4 (north_star's bar).  From step 2
    # on no two fp32 implementations can: Adam's first updates are +-lr per element whatever the gradient's size, and every fp32 gradient is
    # ~1 % from the fp64 one (EM routing amplifies rounding), so every element whose exact gradient is below that noise moves by 2 lr relative to
    # the exact trajectory -- the reference's OWN fp32 run is 2.2e-4 (step 2) and 1.8e-3 (step 3) from its fp64 run on the total loss (1.4e-3 of
    # the class loss's value), and stays there under other thread counts (the t<k>:: runs of the fixture: 1.6 - 1.9e-3).  This engine's arithmetic
    # differs from the reference's structurally (Winograd, row-spectral forms, a double-precision sigma^2 in EM routing), so ITS distance is another
    # draw of the same kind of quantity: measured 1.6e-5 .. 4.7e-4 at step 2 and 1.1e-3 .. 5.3e-3 at step 3 over this round's code states
    # (DESIGN.md 4), and 2 % from run to run (split-K atomics).  The bar: no further from the reference's fp64 trajectory than 3x the
    # reference's fp32 run is, or -- for the class loss and the total that contains it -- 1 % of the class loss's value (7x the reference's own
    # relative distance), whichever is larger.  (Round 4's first
    # version had the 3x term alone: two draws of one distribution fail a 3x-of-one-draw bar a fifth of the time, and this one sat at 0.96 - 1.02
    # of it from run to run.)  The robust statistics of the trajectory are the ones below: running statistics, parameter distances, norms.
    st0 = v["steps"][0]
    assert all(d <= 1e-4 for d in st0["loss_vs_ref32"].values()) and st0["logits_vs_ref32"] <= 1e-3, st0
    for s_, st in enumerate(v["steps"]):
        for k in ("total", "loc", "cls", "cons"):
            rel = 1e-2 * abs(st["loss_f64"]["cls"]) if (s_ > 0 and k in ("cls", "total")) else 0.0      # the class loss carries the noise; total = loc + cls + cons
            assert st["loss_vs_f64"][k] <= max(3 * st["ref32_vs_f64"][k], 1e-4, rel), ("step %d" % s_, k, st)
        assert st["logits_vs_f64"] <= max(3 * st["logits_ref32_vs_f64"], 1e-3), (s_, st)
    for k, d in v["bufs"].items():                          # BatchNorm running statistics after six forward passes
        assert d["vs_f64"] <= max(3 * d["ref32_vs_f64"], 1e-5), (k, d)
    for k, (got, ref) in v["nbt"].items():
        assert got == ref == 6, (k, got, ref)
    for k, d in v["params"].items():                        # parameters after the third Adam step
        assert d["mean_vs_f64"] <= max(3 * d["ref32_mean_vs_f64"], 1e-3 * d["moved"]) and d["max_vs_f64"] <= 2.5 * 3 * 1e-4, (k, d)
        assert d["frac_over_lr"] <= max(3 * d["ref32_frac_over_lr"], 0.01), (k, d)
    # every parameter tensor's norm: distance from the fp64 run beyond 3x the fp32 reference's own, relative to max(norm, 1).  The tensors at the
    # top are BatchNorm biases of the trunk (norm ~ 1, a few hundred elements, each moved by +-lr = 1e-4 per step): one element whose step-3
    # sign differs moves the norm by up to 1e-4 (measured 7.1e-5; 1.19e-4 with Conv3d_2c's forward in Winograd F(4x4, 3x3), which is one of
    # the reasons that launch stays in F(2x2, 3x3): gpurun_out/trajectory_*.json, param_norm_excess_top)
    assert v["param_norm_excess"] <= 1e-4, v.get("param_norm_excess_top")
    # Adam's moments after step 3 (round 5): what guards the optimiser's state machine.  Element-wise, the well-conditioned tensors (decoder,
    # capsule head: 2e-6 .. 1e-2 from the fp64 run) pin the early-Adam range to the reference; the trunk's moments are noise from step 2 on in
    # ANY fp32 arithmetic (the reference's own fp32 run is 40 - 70 % from its fp64 run there, EM routing amplifies rounding), so for every
    # tensor -- the stem, which the closing Adam op owns, included -- there are two norm bars: every tensor's moment norm within a factor 5 of the
    # fp64 run's (or 3x the reference's own fp32 log-distance): a range that an Adam op skipped leaves its tensors' moments at zero or a step
    # behind; and the MEDIAN relative norm distance over all tensors within 3x the reference's (measured 1.7e-2 against 2.0e-2): a doubled or
    # mis-scaled update of either op's range moves half the tensors by tens of percent.
    assert v["adam_step_ref"] == [3, 3] and v["step_count"] == 3
    for k, d in v["adam"].items():
        floor = 2e-3 if k.startswith("adam_m::") else 4e-3
        assert d["vs_f64"] <= max(3 * d["ref32_vs_f64"], floor), (k, d)
    for tag in ("adam_m_norms", "adam_v_norms"):
        assert v[tag]["worst_excess"] <= 0.0, (tag, v[tag])
        assert v[tag]["median_rel"] <= max(3 * v[tag]["ref32_median_rel"], 0.02), (tag, v[tag])
        # the stem's three tensors are the closing Adam op's whole range, and their moments are well conditioned (measured: 0.4 - 8 % from the fp64
        # run, the reference's fp32 run 1.4 - 8 %): a factor 1.25 (or 3x the reference's own log-distance) -- a doubled update doubles them
        assert len(v[tag]["stem"]) == 3
        for name, lg, lr32 in v[tag]["stem"]:
            assert lg <= max(3 * lr32, float(np.log(1.25))), (tag, name, lg, lr32)
