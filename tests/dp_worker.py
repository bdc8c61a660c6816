"""One rank of the data-parallel step test (started twice by tests/test_dp_gpu.py, both ranks on cuda:0, gloo for the
exchange because RCCL refuses two ranks on one device).  Runs the REAL engine path of bench.py at N > 1 --
StepEngine.forward_backward(reducer=GradReducer) with two lanes, segmented backward, bucket all-reduce launched behind the
ops that finalise each bucket, 1/world folded into the fused Adam -- and checks it against

  * two independent single-rank engines on the same device (G_dp == g_0 + g_1, BN running statistics NOT equalised),
  * the mean of two CPU-oracle steps (DP(world x bs) == mean of `world` independent reference steps,
    /root/reference/main_ucf101.py:171-184 is the per-rank loop),
  * the other rank (identical parameters after Adam).

Writes a JSON verdict to argv[1].<rank>; exit code 0 only if every check passed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import picons_amd  # noqa: E402,F401
from picons_amd import dist as pdist, step as pstep, synthetic  # noqa: E402

HW, BS, EPOCH, LR = 112, 2, 1, 1e-4
AKW = dict(bv=True, n_frames=5, wt_cons=0.1)
LANES = 2
# PICONS_DP_FULL=1: the product default -- four lanes, 8x224x224, bs = 8 per rank (what bench.py runs on every rank at N > 1).  Every
# engine-against-engine check runs; the CPU-oracle comparison stays with the 112^2 case (two fp32 + two fp64 oracle steps at bs = 8 take
# minutes, and tests/test_step_gpu.py holds the bs = 8 oracle bars for the single-rank step)
FULL = os.environ.get("PICONS_DP_FULL", "0") == "1"
if FULL:
    HW, BS, LANES = 224, 8, None


def main():
    out_path = sys.argv[1]
    rank, world, _local = pdist.init_from_env(backend="gloo")
    assert world == 2 and dist.get_backend() == "gloo"
    torch.set_num_threads(8)
    res = {"rank": rank, "checks": {}}
    ok = True

    def check(name, cond, info):
        nonlocal ok
        res["checks"][name] = {"ok": bool(cond), "info": info}
        ok = ok and bool(cond)

    ramp = pstep.exp_rampup(100)(EPOCH)
    args = pstep.default_args(lr=LR, **AKW)
    eng = pstep.StepEngine(args, bs=BS, hw=HW, lanes=LANES, device="cuda:0")
    red = eng.make_reducer(target_floats=3_000_000)
    check("host_staged_reducer", red.world == 2 and red.host_staged and len(red.buckets) >= 3, [red.world, len(red.buckets)])
    mine = synthetic.make_step_inputs(BS, rank=rank, step=0, hw=HW)
    eng.stage(*mine)
    eng.forward_backward(EPOCH, ramp, reducer=red)
    red.wait()
    torch.cuda.synchronize()
    G_dp = eng.G.clone()
    R_dp = eng.R.clone()
    scal = eng.read_scalars()

    # the same two minibatches through independent single-rank engines on this device
    solo = pstep.StepEngine(args, bs=BS, hw=HW, lanes=LANES, device="cuda:0")
    g, r_stats = [], []
    for r in range(world):
        solo.load_state(synthetic.init_state(47, 24))
        solo.stage(*synthetic.make_step_inputs(BS, rank=r, step=0, hw=HW))
        solo.forward_backward(EPOCH, ramp)
        torch.cuda.synchronize()
        g.append(solo.G.clone())
        r_stats.append(solo.R.clone())
    gsum = g[0] + g[1]
    rel = ((G_dp - gsum).norm() / gsum.norm()).item()
    check("G_equals_sum_of_rank_gradients", rel < 2e-4, rel)
    # round 6: split-K weight gradients and the merged tail are summed in a fixed order (no fp32 atomics in any gradient), so a
    # rank's gradient is bit-identical from run to run -- and a two-rank fp32 all-reduce is ONE addition per element: the reduced gradient must
    # EQUAL g_0 + g_1 bit for bit
    bad = []
    for name, shp in eng.plan.pshape.items():
        o, n = eng.plan.poff[name], int(np.prod(shp))
        a, b = G_dp[o:o + n], gsum[o:o + n]
        if not torch.equal(a, b):
            bad.append((name, float((a - b).abs().max())))
    check("G_is_bitwise_the_sum_of_rank_gradients", not bad, bad[:6])
    diff01 = ((g[0] - g[1]).norm() / gsum.norm()).item()
    check("rank_gradients_differ", diff01 > 1e-2, diff01)             # the all-reduce had something to do
    check("bn_running_stats_are_per_rank", torch.allclose(R_dp, r_stats[rank], rtol=1e-6, atol=1e-7)
          and not torch.allclose(r_stats[0], r_stats[1], rtol=1e-4, atol=1e-6), float((r_stats[0] - r_stats[1]).abs().max()))

    # Adam with the mean gradient: both ranks end with the same parameters
    eng.adam(LR, red.gscale)
    torch.cuda.synchronize()
    Pn = eng.P.detach().cpu()
    lo, hi = Pn.clone(), Pn.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    check("parameters_identical_on_both_ranks", torch.equal(lo, hi), float((hi - lo).abs().max()))
    ones = torch.ones(1)
    dist.all_reduce(ones)
    check("rank_count_observed", int(ones.item()) == 2, int(ones.item()))

    check("lanes", eng.plan.lanes == (4 if FULL else 2) and len(eng.side) == eng.plan.lanes - 1, eng.plan.lanes)
    if rank == 0 and not FULL:
        # mean of two oracle steps, fp32 and fp64 (the anchor): whole-gradient and per-tensor bars of tests/test_step_gpu.py
        from oracle import step as ostep
        oa = ostep.default_args(**AKW)
        acc = {}
        for dt in (torch.float32, torch.float64):
            tot = None
            for r in range(world):
                P = ostep.as_torch_params(synthetic.init_state(47, 24), dtype=dt)
                lab, unl, perm, drops = synthetic.make_step_inputs(BS, rank=r, step=0, hw=HW)
                ref = ostep.train_step(P, oa, lab, unl, EPOCH, ramp, perm, drops, dtype=dt)
                ref["total"].backward()
                gr = {k: p.grad.double() for k, p in P.items() if p.requires_grad}
                tot = gr if tot is None else {k: tot[k] + gr[k] for k in tot}
                if dt == torch.float32 and r == rank:
                    check("rank0_loss_matches_oracle", abs(scal["total"] - float(ref["total"])) <= 1e-4, [scal["total"], float(ref["total"])])
            acc[dt] = {k: v / world for k, v in tot.items()}
        num_g = num_c = den = 0.0
        bad = []
        for name in eng.plan.pshape:
            o = eng.plan.poff[name]
            n = int(np.prod(eng.plan.pshape[name]))
            mean_hip = (G_dp[o:o + n].cpu().double() * red.gscale).view(eng.plan.pshape[name])
            r64, r32 = acc[torch.float64][name], acc[torch.float32][name]
            d = r64.norm().item() + 1e-12
            rel_g, rel_c = (mean_hip - r64).norm().item() / d, (r32 - r64).norm().item() / d
            num_g += (mean_hip - r64).norm().item() ** 2; num_c += (r32 - r64).norm().item() ** 2; den += d ** 2
            if rel_g > max(4 * rel_c, 5e-3) and (mean_hip - r64).abs().max().item() > 1e-7:
                bad.append([name, rel_g, rel_c])
        tg, tc = (num_g / den) ** 0.5, (num_c / den) ** 0.5
        check("mean_gradient_vs_mean_of_oracle_steps", not bad and tg <= max(3 * tc, 2e-3), {"hip": tg, "fp32_oracle": tc, "bad": bad[:5]})
        # parameters after Adam on the mean gradient
        P = ostep.as_torch_params(synthetic.init_state(47, 24))
        for k, p in P.items():
            if p.requires_grad:
                p.grad = acc[torch.float32][k].float()
        ostep.adam_step(P, {}, {}, 1, LR)
        worst = max((eng.param(k).cpu() - P[k].detach()).abs().max().item() for k in ("conv_caps.beta_u", "smooth.weight", "conv1.Mixed_4f.b0.bn.weight"))
        check("adam_on_mean_gradient", worst <= 5e-5, worst)

    res["ok"] = ok
    with open("%s.%d" % (out_path, rank), "w") as f:
        json.dump(res, f, indent=1)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
