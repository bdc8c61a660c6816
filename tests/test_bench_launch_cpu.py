"""CPU: `python bench.py --gpus N` must become N ranks by itself (the driver's command carries no torchrun), relay exactly one JSON
line whose n_gpus == ranks_observed == N, and fail loudly -- non-zero exit, no line -- when a rank dies or the environment's
world size disagrees with --gpus.  `--dry-launch` keeps the ranks to rendezvous + all-reduce(ones) so this runs without GPUs (gloo)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PICONS_BENCH_FAIL_RANK")}
    e.update(kw)
    return e


def _run(*flags, env=None, timeout=300):
    return subprocess.run([sys.executable, BENCH, *flags], capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env or _env())


def test_gpus_2_starts_two_ranks_and_relays_one_line():
    p = _run("--gpus", "2", "--dry-launch", "--dry-backend", "gloo")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_observed"] == 2 and j["dry_launch"] is True
    assert j["reducer"]["backend"] == "gloo" and j["local_rank"] == 0 and j["master"].startswith("127.0.0.1:")
    assert j["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]


def test_four_ranks():
    p = _run("--gpus", "4", "--dry-launch", "--dry-backend", "gloo")
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0])
    assert j["n_gpus"] == 4 and j["ranks_observed"] == 4


def test_a_dying_rank_fails_the_run_and_stops_the_others():
    t0 = time.time()
    p = _run("--gpus", "2", "--dry-launch", "--dry-backend", "gloo", env=_env(PICONS_BENCH_FAIL_RANK="1"), timeout=120)
    assert p.returncode == 7, (p.returncode, p.stderr[-2000:])
    assert p.stdout.strip() == ""                           # no line for a run that did not happen
    assert "rank 1 exited with code 7" in p.stderr
    assert time.time() - t0 < 90                            # rank 0 was stopped, not left waiting at the rendezvous


def test_world_size_mismatch_is_refused():
    p = _run("--gpus", "2", "--dry-launch", env=_env(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert p.returncode == 2 and p.stdout.strip() == "" and "refusing" in p.stderr
    p = _run("--gpus", "1", "--dry-launch", env=_env(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"))
    assert p.returncode == 2


def test_under_a_torchrun_environment_no_second_launcher():
    """RANK present => bench.py is one rank of somebody else's launch: it must not spawn again."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--dry-launch", "--dry-backend", "gloo"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=_env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))) for r in range(2)]
    outs = [q.communicate(timeout=300) for q in procs]
    assert all(q.returncode == 0 for q in procs), outs
    j = json.loads(outs[0][0].strip())
    assert j["n_gpus"] == 2 and j["ranks_observed"] == 2 and outs[1][0].strip() == ""


def test_launcher_parent_makes_no_gpu_call():
    """The branch that starts the ranks runs before anything that could initialise the GPU: no picons_amd import, no torch.cuda.* call."""
    src = open(BENCH).read()
    body = src[src.index("def launch_ranks"):src.index("def dry_launch")]
    assert "torch.cuda" not in body and "picons_amd" not in body and "os.exec" not in src
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(") < main.index("import picons_amd") and main.index("launch_ranks(") < main.index("torch.cuda")
