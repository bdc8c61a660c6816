"""Worker of tests/test_step_gpu.py::test_training_trajectory_vs_reference: three REAL steps of the fused engine (fresh minibatch each, Adam
between them) against tests/golden/traj_bv5.npz -- the reference's own loop (main_ucf101.py:171-184, Adam of :416).
    python tests/traj_worker.py <default|reducer> <out.json>
default: the product default (four lanes, the early Adam op inside the backward list armed); reducer: the DP schedule through a one-rank
RCCL group (segmented backward, bucket all-reduces, the early Adam behind the launched buckets, 1/world in the optimiser)."""
import ast
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import picons_amd  # noqa: F401
from picons_amd import step as pstep, synthetic

mode, out_path = sys.argv[1], sys.argv[2]
S = np.load(os.path.join(ROOT, "tests", "golden", "traj_bv5.npz"))
bs, nsteps, lr = int(S["bs"]), int(S["nsteps"]), float(S["lr"])
akw = dict(ast.literal_eval(str(S["args"])))
args = pstep.default_args(**{k: v for k, v in akw.items() if k in vars(pstep.default_args())})
args.lr = lr
eng = pstep.StepEngine(args, bs=bs, hw=224, num_classes=24, device="cuda:0")
red = None
if mode == "reducer":
    torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1)
    red = eng.make_reducer(force=True)
res = dict(mode=mode, lanes=eng.plan.lanes, early_adam_op=eng.plan.op_adam_early is not None, steps=[])
F = lambda k: float(S[k])
for s_ in range(nsteps):
    lab, unl, perm, drops = synthetic.make_step_inputs(bs, rank=0, step=int(S["stepids"][s_]), num_classes=24)
    got = eng.train_step(lab, unl, int(S["epoch"]), float(S["ramp"]), perm, drops, lr=lr, reducer=red)
    _out, _flip, pred = eng.outputs()
    eng.synchronize()
    p32, p64 = torch.from_numpy(S["s%d::predicted_action" % s_]), torch.from_numpy(S["f64::s%d::predicted_action" % s_])
    res["steps"].append(dict(
        # distance of this engine and of the reference's own fp32 run from the reference's fp64 run, and the two fp32 runs from each other
        loss_vs_f64={k: abs(got[k] - F("f64::s%d::%s" % (s_, k))) for k in ("total", "loc", "cls", "cons")},
        loss_f64={k: F("f64::s%d::%s" % (s_, k)) for k in ("total", "loc", "cls", "cons")},
        ref32_vs_f64={k: abs(F("s%d::%s" % (s_, k)) - F("f64::s%d::%s" % (s_, k))) for k in ("total", "loc", "cls", "cons")},
        # the reference's fp32 run again under other intra-op thread counts (other reduction orders): informational -- how far its fp32 trajectory
        # is from itself (the bar stays on the one run above)
        ref32_other_threads_vs_f64={k: [abs(F("t%d::s%d::%s" % (t, s_, k)) - F("f64::s%d::%s" % (s_, k))) for t in (S["spread_threads"].tolist() if "spread_threads" in S.files else [])]
                                    for k in ("total", "loc", "cls", "cons")},
        loss_vs_ref32={k: abs(got[k] - F("s%d::%s" % (s_, k))) for k in ("total", "loc", "cls", "cons")},
        logits_vs_f64=float((pred.cpu().double() - p64).abs().max()), logits_ref32_vs_f64=float((p32.double() - p64).abs().max()),
        logits_vs_ref32=float((pred.cpu() - p32).abs().max()), total=got["total"]))
sd = eng.state_dict()
res["bufs"], res["params"], res["nbt"] = {}, {}, {}
state0 = synthetic.init_state(seed=47, num_classes=24, conditioned=True)
for k in S.files:
    if k.startswith("buf::"):
        if k.endswith("num_batches_tracked"):
            res["nbt"][k[5:]] = [int(sd[k[5:]]), int(S[k])]
        else:
            r64 = torch.from_numpy(S["f64::" + k])
            res["bufs"][k[5:]] = dict(vs_f64=float((sd[k[5:]].cpu().double() - r64).abs().max()), ref32_vs_f64=float((torch.from_numpy(S[k]).double() - r64).abs().max()),
                                      scale=float(r64.abs().max()))
    if k.startswith("param::"):
        got = sd[k[7:]].cpu().double().numpy(); r32 = S[k].astype(np.float64); r64 = S["f64::" + k]; init = np.asarray(state0[k[7:]], np.float64)
        res["params"][k[7:]] = dict(mean_vs_f64=float(np.abs(got - r64).mean()), ref32_mean_vs_f64=float(np.abs(r32 - r64).mean()), moved=float(np.abs(r64 - init).mean()),
                                    max_vs_f64=float(np.abs(got - r64).max()), frac_over_lr=float((np.abs(got - r64) > lr).mean()),
                                    ref32_frac_over_lr=float((np.abs(r32 - r64) > lr).mean()))
names = [str(x) for x in S["param_names"]]
# the optimiser's state after the last step: Adam's moments are smooth in the gradients -- a wrong step count, a doubled or a
# skipped (early-)Adam range moves them by tens of percent, where the loss scalars above ride on EM routing's noise
res["adam"] = {}
flat_of = lambda buf, n: buf[eng.plan.poff[n]:eng.plan.poff[n] + int(np.prod(eng.plan.pshape[n]))].view(eng.plan.pshape[n]).cpu().double().numpy()
rel = lambda a, b: float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))
for k in S.files:
    if k.startswith("adam_m::") or k.startswith("adam_v::"):
        n, buf = k[8:], (eng.M if k.startswith("adam_m::") else eng.V)
        r32, r64 = S[k].astype(np.float64), S["f64::" + k]
        res["adam"][k] = dict(vs_f64=rel(flat_of(buf, n), r64), ref32_vs_f64=rel(r32, r64))
for tag, buf in (("adam_m_norms", eng.M), ("adam_v_norms", eng.V)):
    got = np.array([float(np.linalg.norm(flat_of(buf, n).ravel())) for n in names])
    r32, r64 = S[tag].astype(np.float64), S["f64::" + tag]
    # per tensor: log-ratio of the moment's norm to the fp64 run's, beside the reference's own fp32 run's (from step 2 on the trunk's
    # gradients are noise through EM routing -- the reference's fp32 moments are 40 - 70 % from its fp64 ones ELEMENT-wise there -- but a
    # tensor's norm mostly stays within tens of percent: measured worst case x2.0 for the first and x4.2 for the second moment of
    # primary_caps.a.bias, whose exact gradient nearly cancels; the bar is a factor 5)
    lg = np.abs(np.log(np.maximum(got, 1e-300) / np.maximum(r64, 1e-300)))
    lr32 = np.abs(np.log(np.maximum(r32, 1e-300) / np.maximum(r64, 1e-300)))
    # (round 5: x6.2 with Conv3d_2c's forward in Winograd F(4x4, 3x3) -- one of the reasons that launch stays in F(2x2, 3x3))
    ex = lg - np.maximum(3 * lr32, np.log(5.0))
    i = int(np.argmax(ex))
    order = np.argsort(-lg)[:5]
    res[tag] = dict(worst_excess=float(ex[i]), worst=names[i], worst_log_ratio=float(lg[i]), worst_ref32_log_ratio=float(lr32[i]),
                    largest=[(names[q], float(lg[q]), float(lr32[q])) for q in order],
                    stem=[(n, float(lg[names.index(n)]), float(lr32[names.index(n)])) for n in names if n.startswith("conv1.Conv3d_1a_7x7.")],
                    max_log_ratio=float(lg.max()), ref32_max_log_ratio=float(lr32.max()),
                    median_rel=float(np.median(np.abs(got - r64) / np.maximum(r64, 1e-30))),
                    ref32_median_rel=float(np.median(np.abs(r32 - r64) / np.maximum(r64, 1e-30))))
res["adam_step_ref"] = [int(S["adam_step"].min()), int(S["adam_step"].max())]
norms = {n: float(sd[n].double().norm()) for n in names}
# every parameter tensor's norm after the last step: distance from the fp64 run beyond 3x the reference's own fp32 distance, relative to max(norm, 1)
# (biases start at zero: after three steps their norm IS the handful of +-lr moves, sign noise included)
excess = sorted(((abs(norms[n] - float(r64)) - 3 * abs(float(r32) - float(r64))) / max(float(r64), 1.0), n, abs(norms[n] - float(r64)), abs(float(r32) - float(r64)), float(r64))
                for n, r32, r64 in zip(names, S["param_norms"], S["f64::param_norms"]))[::-1]
res["param_norm_excess"] = excess[0][0]
res["param_norm_excess_top"] = [dict(excess=e, name=n, d_vs_f64=a, ref32_d_vs_f64=b, norm=c) for e, n, a, b, c in excess[:6]]
res["step_count"] = eng.step_count
json.dump(res, open(out_path, "w"), indent=1)
if red is not None:
    torch.distributed.destroy_process_group()
