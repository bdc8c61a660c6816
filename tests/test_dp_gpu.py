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


def test_two_rank_dp_step_on_one_gpu(tmp_path):
    port = _free_port()
    out = str(tmp_path / "dp")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
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
    with open(os.path.join(ROOT, "gpurun_out", "dp_two_rank_checks.json"), "w") as f:
        json.dump(verdicts, f, indent=1)
    for rank, v in enumerate(verdicts):
        failed = {k: c for k, c in v["checks"].items() if not c["ok"]}
        assert not failed and procs[rank].returncode == 0, (rank, failed, logs[rank][-2000:])
    assert "mean_gradient_vs_mean_of_oracle_steps" in verdicts[0]["checks"]
