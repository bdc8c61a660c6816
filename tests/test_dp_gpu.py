"""Data parallelism with the real engine on ONE GPU: two processes share cuda:0, each runs
StepEngine.forward_backward(reducer=GradReducer) with two lanes on its own minibatch, the gradient buckets are exchanged
over gloo (RCCL cannot put two ranks on one device), then the fused Adam with 1/world.  tests/dp_worker.py holds the checks
(sum of rank gradients, mean of two oracle steps, per-rank BN statistics, identical parameters afterwards)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("full", [False, True])
def test_two_rank_dp_step_on_one_gpu(tmp_path, full):
    """full=True: the product default (four lanes, 8x224x224, bs = 8 per rank) -- every engine-against-engine check; the oracle
    comparison runs in the small case."""
    port = _free_port()
    out = str(tmp_path / "dp")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PICONS_DP_FULL="1" if full else "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    verdicts = []
    for rank in range(2):
        path = "%s.%d" % (out, rank)
        assert os.path.exists(path), "rank %d wrote no verdict:\n%s" % (rank, logs[rank][-3000:])
        verdicts.append(json.load(open(path)))
    with open(os.path.join(ROOT, "gpurun_out", "dp_two_rank_checks%s.json" % ("_full" if full else "")), "w") as f:
        json.dump(verdicts, f, indent=1)
    for rank, v in enumerate(verdicts):
        failed = {k: c for k, c in v["checks"].items() if not c["ok"]}
        assert not failed and procs[rank].returncode == 0, (rank, failed, logs[rank][-2000:])
    assert full or "mean_gradient_vs_mean_of_oracle_steps" in verdicts[0]["checks"]
    assert verdicts[0]["checks"]["lanes"]["info"] == (4 if full else 2)


RCCL_ONE_RANK = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
import picons_amd
from picons_amd import step as pstep, synthetic
dist.init_process_group(backend="nccl", rank=0, world_size=1)
args = pstep.default_args(lr=1e-4, bv=True, n_frames=5, wt_cons=0.1)
eng = pstep.StepEngine(args, bs=2, hw=112, device="cuda:0")
eng.stage(*synthetic.make_step_inputs(2, rank=0, step=0, hw=112))
ramp = pstep.exp_rampup(100)(1)
eng.forward_backward(1, ramp)
torch.cuda.synchronize()
G0 = eng.G.clone()
red = eng.make_reducer(target_floats=3_000_000, force=True)
assert red.active and not red.host_staged and red.world == 1 and dist.get_backend() == "nccl"
eng.load_state(synthetic.init_state(47, 24))
eng.forward_backward(1, ramp, reducer=red)
n_launched = len(red.launched)
red.wait()
torch.cuda.synchronize()
rel = ((eng.G - G0).norm() / G0.norm()).item()
ones = torch.ones(4, device="cuda:0")
dist.all_reduce(ones)
out = eng.run_staged(1, ramp, reducer=red)          # the whole DP step: segmented backward, RCCL buckets, Adam with gscale
torch.cuda.synchronize()
json.dump({"rel": rel, "buckets": len(red.buckets), "launched": n_launched, "ones": ones.tolist(), "gscale": red.gscale,
           "total": out["total"]}, open(sys.argv[1], "w"))
dist.destroy_process_group()
"""


def test_rccl_one_rank_group_runs_the_dp_schedule(tmp_path):
    """The N > 1 code path through RCCL itself on the one GPU a box has: a one-rank "nccl" process group, GradReducer forced
    active, the backward replayed in bucket segments with every bucket's all-reduce issued on the comm stream behind the lanes'
    events.  An all-reduce over one rank is the identity, so the gradient must equal the unsegmented one (fp32 split-K atomics
    reorder sums only)."""
    out = str(tmp_path / "rccl.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", RCCL_ONE_RANK % {"root": ROOT}, out], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and os.path.exists(out), p.stdout[-3000:]
    v = json.load(open(out))
    with open(os.path.join(ROOT, "gpurun_out", "rccl_one_rank.json"), "w") as f:
        json.dump(v, f)
    assert v["buckets"] >= 3 and v["launched"] == v["buckets"], v
    assert v["rel"] < 2e-4 and v["ones"] == [1.0] * 4 and v["gscale"] == 1.0, v
    assert v["total"] == v["total"], v


EARLY_ADAM_DP = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
import picons_amd
from picons_amd import step as pstep, synthetic
dist.init_process_group(backend="nccl", rank=0, world_size=1)
args = pstep.default_args(lr=1e-4, bv=True, n_frames=5, wt_cons=0.1)
ramp = pstep.exp_rampup(100)(1)
res = {}
P = {}
for tag, flag in (("early", "1"), ("late", "0")):
    os.environ["PICONS_EARLY_ADAM_DP"] = flag
    eng = pstep.StepEngine(args, bs=2, hw=112, device="cuda:0")
    P0 = eng.P.clone()
    red = eng.make_reducer(target_floats=3_000_000, force=True)
    eng.stage(*synthetic.make_step_inputs(2, rank=0, step=0, hw=112))
    out = eng.run_staged(1, ramp, reducer=red)
    eng.synchronize()
    torch.cuda.synchronize()
    P[tag] = eng.P.clone()
    res[tag] = {"final_adam_split": eng.last_adam_split, "nparams": eng.plan.nparams, "last_bucket_end": sorted(red.buckets)[-1][2],
                "moved": ((P[tag] - P0).abs() > 0).float().mean().item(), "total": out["total"], "step_count": eng.step_count}
d = (P["early"] - P["late"])
upd = (P["late"] - P0)
res["rel_update_diff"] = (d.norm() / upd.norm()).item()
res["frac_differ"] = (d.abs() > 1e-9).float().mean().item()
json.dump(res, open(sys.argv[1], "w"))
dist.destroy_process_group()
"""


def test_early_adam_under_the_reducer_matches_the_late_adam(tmp_path):
    """VERDICT r3 #8c: under data parallelism the early Adam op covers the suffix of the flat buffer whose buckets have been launched when the
    backward reaches it (everything but the trunk's last bucket), behind those collectives on the device; the final Adam takes the rest.  One
    step through a one-rank RCCL group with the split optimiser against the same step with PICONS_EARLY_ADAM_DP=0 (one Adam behind
    reducer.wait()): every parameter moved exactly once, and the two parameter sets agree up to the run-to-run noise of the split-K atomics
    (Adam's first update is lr * sign(g) wherever |g| >> eps, so they agree far better than the gradients do)."""
    out = str(tmp_path / "early_dp.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", EARLY_ADAM_DP % {"root": ROOT}, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert p.returncode == 0 and os.path.exists(out), p.stdout[-3000:]
    v = json.load(open(out))
    with open(os.path.join(ROOT, "gpurun_out", "early_adam_dp.json"), "w") as f:
        json.dump(v, f)
    e, l = v["early"], v["late"]
    # the early op took everything behind the bucket that leaves last (the head of the flat buffer: the trunk's first layers) ...
    assert e["final_adam_split"] == e["last_bucket_end"] and 0 < e["final_adam_split"] < e["nparams"] // 4, v
    assert l["final_adam_split"] == l["nparams"], v
    # ... and nothing was stepped twice or not at all: the same elements move as with one Adam (a parameter whose gradient is exactly zero -- a
    # class capsule no sample of the minibatch belongs to -- does not move in either)
    assert e["step_count"] == l["step_count"] == 1 and e["moved"] > 0.5 and abs(e["moved"] - l["moved"]) < 1e-4, v
    assert v["rel_update_diff"] < 2e-2 and v["frac_differ"] < 2e-2, v
    assert abs(e["total"] - l["total"]) <= 1e-6 * abs(l["total"]), v
