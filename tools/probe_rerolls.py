#!/usr/bin/env python3
"""How much of a tensor's distance from the fp64 oracle is a draw?  One bs = 8, 8x224x224 case of tests/test_step_gpu.py (BS8), the CPU oracle
(fp32 and fp64) computed ONCE, and the HIP step run under several settings that are all fp32-accurate but round differently (another
kernel form for a layer, another summation order): per setting the whole-gradient rel-L2, the tensors over the test's bar
max(4 x fp32-oracle, 5e-3), the largest bar ratios and the named tensors.

    python tools/probe_rerolls.py <case index> <tensor,tensor,...> name:ENV=v,ENV=v [name:ENV=v ...]
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
CACHE = os.environ.get("PICONS_REROLL_CACHE", "/tmp/picons_reroll_oracle.pt")


def child(case, names):
    import torch
    import test_step_gpu as T
    from picons_amd import step as pstep, synthetic
    tag, akw, stepid, epoch, ncls, jhmdb = T.BS8[case]
    args = pstep.default_args(lr=1e-4, **akw)
    eng = pstep.StepEngine(args, bs=8, hw=224, num_classes=ncls, jhmdb=jhmdb, state=synthetic.init_state(47, ncls))
    lab, unl, perm, drops = synthetic.make_step_inputs(8, rank=0, step=stepid, num_classes=ncls, hw=224)
    eng.stage(lab, unl, perm, drops)
    eng.forward_backward(epoch, pstep.exp_rampup(100)(epoch))
    torch.cuda.synchronize()
    O = torch.load(CACHE)
    rows, ng, nc, da = [], 0.0, 0.0, 0.0
    for name in eng.plan.pshape:
        g = eng.grad(name).cpu().double(); r32, r64 = O[name]
        den = r64.norm().item() + 1e-12
        eg, ec = (g - r64).norm().item(), (r32.double() - r64).norm().item()
        ng += eg ** 2; nc += ec ** 2; da += den ** 2
        rows.append((eg / den / max(4 * ec / den, 5e-3), name, eg / den, ec / den))
    rows.sort(reverse=True)
    print("  whole-gradient rel-L2: hip %.3e  fp32 oracle %.3e;  tensors over the bar: %d of %d" % ((ng / da) ** 0.5, (nc / da) ** 0.5, sum(r[0] > 1 for r in rows), len(rows)))
    for r in rows[:4]:
        print("    %.2f of its bar  %-44s hip %.3e  fp32 oracle %.3e" % r)
    for r in rows:
        if r[1] in names:
            print("    named: %.2f of its bar  %-37s hip %.3e  fp32 oracle %.3e" % r)


def main():
    case, names, variants = int(sys.argv[1]), sys.argv[2].split(","), sys.argv[3:]
    if os.environ.get("PICONS_REROLL_CHILD"):
        return child(case, names)
    import torch
    import test_step_gpu as T
    from oracle import step as ostep
    from picons_amd import step as pstep, synthetic
    tag, akw, stepid, epoch, ncls, jhmdb = T.BS8[case]
    if not os.path.exists(CACHE):
        state = synthetic.init_state(47, ncls)
        lab, unl, perm, drops = synthetic.make_step_inputs(8, rank=0, step=stepid, num_classes=ncls, hw=224)
        ramp = pstep.exp_rampup(100)(epoch)
        oa = ostep.default_args(dataset="jhmdb" if jhmdb else "ucf101", **akw)
        P = ostep.as_torch_params(state)
        ostep.train_step(P, oa, lab, unl, epoch, ramp, perm, drops)["total"].backward()
        P64 = ostep.as_torch_params(state, dtype=torch.float64)
        ostep.train_step(P64, oa, lab, unl, epoch, ramp, perm, drops, dtype=torch.float64)["total"].backward()
        torch.save({k: (P[k].grad, P64[k].grad) for k in P if P[k].grad is not None}, CACHE)
    print("case", tag)
    for v in variants:
        label, _, envs = v.partition(":")
        env = dict(os.environ, PICONS_REROLL_CHILD="1")
        for kv in filter(None, envs.split(",")):
            k, _, val = kv.partition("=")
            env[k] = val
        print("== %s (%s)" % (label, envs or "default"), flush=True)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), str(case), ",".join(names)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print("\n".join(l for l in p.stdout.splitlines() if l.startswith("  ")) or p.stdout[-1500:], flush=True)


if __name__ == "__main__":
    main()
